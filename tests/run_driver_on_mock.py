"""One rank of bench.py or bench_workloads.py on the plaintext stand-in (tests/mock_engine.py), CPU
tensors and gloo: `python tests/run_driver_on_mock.py bench|workloads [corrupt_rank] -- <driver flags>`.
Started as fresh processes (RANK / WORLD_SIZE / MASTER_* in the environment) by tests/test_host_logic.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests.mock_engine import MockBenchBackend  # noqa: E402


def main():
    which = sys.argv[1]
    sep = sys.argv.index("--")
    corrupt = int(sys.argv[2]) if sep > 2 else None
    flags = sys.argv[sep + 1:]
    backend = MockBenchBackend(corrupt_rank=corrupt)
    if which == "bench":
        import bench
        bench.main(flags, backend=backend)
    else:
        import bench_workloads
        bench_workloads.main(flags, backend=backend)


if __name__ == "__main__":
    main()
