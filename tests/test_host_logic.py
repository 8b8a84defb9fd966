"""CPU-side checks: the C ABI library loads and exports every declared entry point, error paths
that need no GPU, and the multi-process layout (gloo, two ranks) of the sharded path."""
import ctypes
import os
import re
import socket

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from fhe_amd import native
    hdr = open(os.path.join(ROOT, "include", "lfhe.h")).read()
    declared = sorted(set(re.findall(r"\b(lfhe_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 36
    lib = ctypes.CDLL(native.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/lfhe.h but not exported"
    L = native.load()
    assert sorted(L._declared) == declared, "ctypes binding and header disagree"
    assert L.lfhe_version().startswith(b"lfhe")


def test_no_cpu_fallback_without_gpu():
    import torch
    from fhe_amd import native
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(native.LfheError):
        native.Context()


def test_context_free_entry_points():
    from fhe_amd import native
    L = native.load()
    p = native.Params()
    L.lfhe_default_params(ctypes.byref(p))
    assert (p.lwe_dim, p.poly_size, p.pbs_base_log, p.ks_level) == (918, 2048, 23, 4)
    assert L.lfhe_op_from_name(b"add") == 0 and L.lfhe_op_from_name(b"select") == 16
    assert L.lfhe_op_from_name(b"nope") == -1
    assert L.lfhe_ctx_create(None, 1, 0, bytes(32)) == -1          # LFHE_ERR_ARG
    h = ctypes.c_void_p()
    assert L.lfhe_ctx_create(ctypes.byref(h), 99, 0, bytes(32)) == -5  # unknown parameter set
    assert L.lfhe_last_error(None) == b"null context"


def test_arena_chunk_respects_memory_and_the_grid_limit():
    """Instances per pass of a batched operator: bounded by the arena budget AND by the 65535
    instances one scatter / gather launch can index, so n > 65535 runs in passes instead of failing
    at launch."""
    from fhe_amd import native
    L = native.load()
    blk = 2049 * 8
    assert L.lfhe_arena_chunk(100 * 40 * blk, 40 * blk) == 100
    assert L.lfhe_arena_chunk(0, 40 * blk) == 1                      # always at least one instance
    assert L.lfhe_arena_chunk(1 << 37, 4 * blk) == 65535             # 128 GiB would hold two million euint8
    n = 200_000
    cap = L.lfhe_arena_chunk(1 << 37, 4 * blk)
    passes = [(off, min(cap, n - off)) for off in range(0, n, cap)]
    assert len(passes) == 4 and sum(c for _, c in passes) == n and max(c for _, c in passes) <= 65535


def test_shard_bounds_cover_everything():
    from fhe_amd.sharding import shard_bounds
    for total in (0, 1, 7, 4096, 10000):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(total, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from fhe_amd.sharding import shard_bounds, replicate, replicate_blob, max_over_ranks
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # key images: rank 0 holds them, the others receive into preallocated buffers
    g = torch.Generator().manual_seed(7)
    bsk = torch.randint(0, 256, (1 << 16,), dtype=torch.uint8, generator=g) if rank == 0 else torch.zeros(1 << 16, dtype=torch.uint8)
    ksk = torch.randint(0, 256, (1 << 12,), dtype=torch.uint8, generator=g) if rank == 0 else torch.zeros(1 << 12, dtype=torch.uint8)
    replicate(dist, [bsk, ksk])
    blob = replicate_blob(dist, b"client-key" if rank == 0 else None)
    # data path: each rank works on its own slice, no collective; results gathered by the host
    lo, hi = shard_bounds(1001, world, rank)
    part = [(7 * i + 3) % 16 for i in range(lo, hi)]
    t = max_over_ranks(dist, 1.0 + rank)
    q.put((rank, int(bsk.sum()), int(ksk.sum()), blob, (lo, hi), part, t))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_layout_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, b0, k0, blob0, cut0, part0, t0), (r1, b1, k1, blob1, cut1, part1, t1) = got
    assert (b0, k0) == (b1, k1) and b0 > 0          # both ranks hold the same key images
    assert blob0 == blob1 == b"client-key"
    assert cut0 == (0, 501) and cut1 == (501, 1001)
    assert part0 + part1 == [(7 * i + 3) % 16 for i in range(1001)]
    assert t0 == t1 == 2.0                          # max over ranks


def _workload_rank(rank, world, port, q):
    """Sharded transfers and the sharded argmax on the plaintext stand-in, gloo."""
    import random
    import sys
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from fhe_amd import workloads
    from fhe_amd.sharding import shard_bounds
    from tests.mock_engine import MockBackend
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B = MockBackend()
    rnd = random.Random(5)
    n = 37                                              # ragged: slices of 19 and 18
    bids = rnd.sample(range(1 << 32), n)
    lo, hi = shard_bounds(n, world, rank)
    vals = [B.encrypt(v, 32) if lo <= i < hi else None for i, v in enumerate(bids)]
    idxs = [B.encrypt(i, 16) if lo <= i < hi else None for i in range(n)]
    win = workloads.sharded_argmax(dist, B, vals, idxs, rank, world)
    got = None if win is None else (B.decrypt(win[0]), B.decrypt(win[1]))
    frm = [rnd.randrange(1000) for _ in range(n)]
    to = [rnd.randrange(1000) for _ in range(n)]
    amt = [rnd.randrange(1500) for _ in range(n)]
    enc = lambda vs: [B.encrypt(v, 64) for v in vs[lo:hi]]
    nf, nt = workloads.transfers(B, enc(frm), enc(to), enc(amt), B.encrypt(0, 64))
    q.put((rank, got, (max(bids), bids.index(max(bids))), [B.decrypt(c) for c in nf], [B.decrypt(c) for c in nt],
           frm, to, amt, (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_workloads_two_ranks_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_workload_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][1] == got[0][2] and got[1][1] is None      # rank 0 holds the winner
    for rank, _, _, nf, nt, frm, to, amt, (lo, hi) in got:
        moved = [a if a <= f else 0 for a, f in zip(amt[lo:hi], frm[lo:hi])]
        assert nf == [f - v for f, v in zip(frm[lo:hi], moved)]
        assert nt == [t + v for t, v in zip(to[lo:hi], moved)]
    # one rank, no process group: the same functions
    from fhe_amd import workloads
    from tests.mock_engine import MockBackend
    B = MockBackend()
    w = workloads.sharded_argmax(None, B, [B.encrypt(v, 32) for v in (5, 9, 2)], [B.encrypt(i, 16) for i in range(3)], 0, 1)
    assert (B.decrypt(w[0]), B.decrypt(w[1])) == (9, 1)
    assert workloads.transfers(B, [], [], [], B.encrypt(0, 64)) == ([], [])


def _run_driver(which, world, flags, corrupt_rank=None, timeout=240):
    """bench.py / bench_workloads.py as `world` fresh processes on the stand-in; returns
    (exit codes, stdout of rank 0)."""
    import subprocess
    import sys
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        cmd = [sys.executable, os.path.join(ROOT, "tests", "run_driver_on_mock.py"), which]
        if corrupt_rank is not None:
            cmd.append(str(corrupt_rank))
        procs.append(subprocess.Popen(cmd + ["--"] + flags, env=env, cwd=ROOT, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return [p.returncode for p in procs], outs[0][0], [o[1] for o in outs]


def test_bench_control_flow_two_ranks_gloo():
    """bench.py --gpus 2 end to end on the stand-in: key images broadcast from rank 0 and attached
    on rank 1, each rank bootstraps its own blocks, barrier + maximum over ranks around the timed
    region, verification flag reduced over ranks, ONE JSON line from rank 0 with whole-job
    throughput; then a rank whose outputs are wrong makes EVERY rank exit non-zero."""
    import json
    flags = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "64", "--no-cpu-baseline"]
    codes, out, errs = _run_driver("bench", 2, flags)
    assert codes == [0, 0], errs
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["verified"] is True
    assert d["config"]["batch_per_gpu"] == 64 and "x2" in d["config"]["parallelism"]
    assert abs(d["value"] - 64 * 2 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]      # all ranks' blocks
    assert {a["flavour"] for a in d["other_bootstrapping_keys"]} == {"q64", "q64_grouped", "q64_grouped3", "prime", "grouped"} and "value_q64_key" in d and "value_q64_grouped_key" in d
    assert d["value_param_set_as_published"]["own_key_with_the_key_bits_in_pairs"] == d["value_q64_grouped_key"]
    assert d["value_param_set_as_published"]["own_key_with_the_key_bits_in_triples"] == d["value_q64_grouped3_key"]
    assert d["value_param_set_as_published"]["euint64_mul_latency_ms_with_the_key_bits_in_triples"] > 0
    assert d["mul_pbs"] > 0 and "cpu_baseline" not in d and d["roofline"]["traffic"] is None
    codes, out, errs = _run_driver("bench", 2, flags + ["--no-alt", "--no-operators"], corrupt_rank=1)
    assert all(c != 0 for c in codes), (codes, errs)
    assert '"verified": false' in out
    # one rank, no process group
    codes, out, errs = _run_driver("bench", 1, ["--steps", "1", "--warmup", "0", "--batch", "32", "--no-cpu-baseline",
                                               "--no-alt"])
    assert codes == [0] and json.loads(out.splitlines()[-1])["n_gpus"] == 1, errs
    # a launch whose world size contradicts --gpus refuses to run
    codes, _, _ = _run_driver("bench", 1, ["--gpus", "2"])
    assert codes == [1]


def test_workload_driver_two_ranks_gloo():
    """bench_workloads.py --gpus 2 on the stand-in: instances sliced per rank, bootstrap counts
    summed over ranks, the argmax winners gathered and reduced on rank 0, every decrypted result
    checked on the rank that owns it."""
    import json
    flags = ["--gpus", "2", "--transfers", "10", "--bids", "21", "--mul-batch", "5"]
    codes, out, errs = _run_driver("workloads", 2, flags)
    assert codes == [0, 0], errs
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["erc20"]["transfers"] == 10 and d["argmax"]["bids"] == 21
    assert d["euint64"]["mul_batch"] == 5 + 6          # the reference's six vectors ride along
    assert d["erc20"]["pbs"] > 0 and d["argmax"]["pbs"] > 0


def test_lazy_reduction_schedule_stays_exact():
    """The per-register normalisation masks compiled into the kernels keep every lazily reduced
    value below 2^53 in the worst case (bound model: tools/lazy_bounds.py) and hand the
    accumulator update a value below 2 p."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("lazy_bounds", os.path.join(root, "tools", "lazy_bounds.py"))
    lb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lb)
    worst, final, norms = lb.check_header(os.path.join(root, "fhe_amd", "csrc", "pbs_device.h"))
    assert worst < lb.LIMIT and final <= 2.0 and norms == 14
    # the earlier schedule (everything after stages 2, 6, 10) does not pass this model
    allr = set(range(8))
    w_old, _, n_old = lb.inverse({(0, 1): allr, (2, 0): allr, (3, 1): allr}, 2 * lb.prod(lb.forward()[-1]))
    assert w_old > lb.LIMIT and n_old == 24


def test_typescript_surface_matches_the_binding_and_the_reference_class():
    """bindings/node/index.d.ts declares exactly what index.js exports, and LuxFHEClient.ts carries
    every public method of the reference class (packages/luxfhejs/src/index.ts:42-200)."""
    import json
    import shutil
    import subprocess
    node_dir = os.path.join(ROOT, "bindings", "node")
    dts = open(os.path.join(node_dir, "index.d.ts")).read()
    ts = open(os.path.join(node_dir, "LuxFHEClient.ts")).read()
    reference_methods = ["initialize", "getPublicKey", "encrypt_uint8", "encrypt_uint16", "encrypt_uint32", "encrypt_uint64",
                         "encrypt_uint128", "encrypt_uint256", "encrypt_address", "unseal", "extractPermitPermission",
                         "storePermit", "getStoredPermit"]
    for m in reference_methods:
        assert re.search(r"^\s+(async\s+)?%s\(" % m, ts, re.M), f"LuxFHEClient.ts lacks {m}"
    assert "serverUrl" in ts and "provider" in ts           # the reference's config fields
    if not shutil.which("node") or not os.path.exists(os.path.join(node_dir, "lfhe_napi.node")):
        pytest.skip("node or the built addon is not here")
    js = "const m=require('./index.js');console.log(JSON.stringify({e:Object.keys(m)," \
         "n:Object.getOwnPropertyNames(m.NativeEngine.prototype),c:Object.getOwnPropertyNames(m.LuxFHEClient.prototype)}))"
    got = json.loads(subprocess.run(["node", "-e", js], cwd=node_dir, capture_output=True, text=True, check=True).stdout)
    for name in got["e"]:
        assert re.search(r"export (declare )?(class|const|function) %s\b" % name, dts), f"index.d.ts lacks export {name}"
    block = lambda cls: re.search(r"export class %s \{(.*?)\n\}" % cls, dts, re.S).group(1)
    for cls, names in (("NativeEngine", got["n"]), ("LuxFHEClient", got["c"])):
        declared = set(re.findall(r"^\s+(?:readonly )?(\w+)[(:]", block(cls), re.M))
        assert set(names) - {"constructor"} <= declared, (cls, set(names) - declared)
        assert declared - set(names) <= {"engine", "constructor"}, (cls, declared - set(names))


def test_blind_rotation_kernels_keep_their_registers():
    """The blind-rotation kernels the library launches hold their state in registers: no scratch (spills turn
    the FP64-bound kernels memory-bound), at most 256 VGPRs, two waves per SIMD.  Read from the compiler's
    resource remarks that the build keeps next to the objects (the slow path of the lift guard, which never
    runs, is allowed to spill)."""
    import re
    path = os.path.join(ROOT, "fhe_amd", "_build", "kernels.hip.o.resources.txt")
    src = os.path.join(ROOT, "fhe_amd", "csrc", "kernels.hip")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        pytest.skip("no resource remarks for the current kernels.hip (python fhe_amd/build.py writes them)")
    kernels, name = {}, None
    for line in open(path):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            kernels[name] = {}
            continue
        m = re.search(r" (VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and name:
            kernels[name][m.group(1).split(" ")[0]] = int(m.group(2))
    hot = {k: v for k, v in kernels.items() if "k_blind_rotate" in k}
    assert len(hot) >= 20, sorted(kernels)
    # mangled template arguments of k_blind_rotate_g64_pair<FACTORED, NC, SAFE, BITS>: ...Lb<F>ELi<NC>ELb<SAFE>ELi<BITS>E
    slow_path = [k for k in hot if re.search(r"g64_pairILb[01]ELi1ELb1E", k)]
    assert len(slow_path) == 3
    for k, v in hot.items():
        assert v["VGPRs"] <= 256, (k, v)
        if k in slow_path:
            continue
        assert v["ScratchSize"] == 0, (k, v)
        assert v["Occupancy"] >= 2 or "wide" in k, (k, v)      # the 512-thread kernels: one workgroup of eight waves per CU
