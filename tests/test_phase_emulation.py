"""The device phase code (fhe_amd/csrc/pbs_phases.h) compiled for the host and run for 256 emulated
threads against the oracle, under the most hostile wavefront schedule the kernels' synchronisation
allows: between two workgroup barriers each wavefront runs ahead on its own, in a scrambled order.
Bit-exact results there show that every hand-over the kernels protect with a wave-level fence only
is in fact private to one wavefront."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_phases_under_hostile_wave_schedule(tmp_path):
    from oracle.oracle import build
    build()
    exe = tmp_path / "ep_emu"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", os.path.join(ROOT, "tests/emu/ep_emu.cpp"),
                    "-o", str(exe), "-L" + os.path.join(ROOT, "oracle"), "-loracle",
                    "-Wl,-rpath," + os.path.join(ROOT, "oracle")], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("OK")
    assert r.stdout.count("mismatches: 0 ") == 8
