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


def test_lift_guard_marks_the_end_of_the_window_and_a_forced_overflow(tmp_path):
    """crt_wrap_checked, the device code, on the host: integers well inside the lift's window come back
    exact and unmarked, integers in the guard band exact and marked, integers beyond the window wrong
    (by construction of a lift) and marked -- the mark is what re-runs their ciphertext on the slow,
    unconditionally exact path of the grouped key over 2^64."""
    exe = tmp_path / "lift_guard"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", os.path.join(ROOT, "tests/emu/lift_guard.cpp"),
                    "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
