"""The frame error of the visual update is a SERIAL float accumulation in the reference (src/vio.cpp:1554, 1634: `error += patch_error` inside an OpenMP static
block), and its exact bits decide accept / revert (vio.cpp:1648).  fast-livo2_amd/csrc/float_chain.hpp evaluates that chain on whole waves; tools/float_chain_model.cpp
is the lane-by-lane CPU model of that code (guesses, one plain round, start -> end tables on eight consecutive
floats per lane, index maps composed in a scan).  Here the model is fuzzed against the serial loop: similar-sized errors, wide ranges, zeros, leading zeros, short mantissas
(a tie on most adds of some binades), subnormals, constants, negative / infinite / NaN elements, 16 / 32 / 64 lanes per chain, one and several passes per chain.
The device code itself is held against the same serial loop in tests/test_float_chain_gpu.py."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_model_equals_the_serial_float_loop(tmp_path):
    exe = str(tmp_path / "float_chain_model")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tools", "float_chain_model.cpp")], check=True)
    out = subprocess.run([exe, "6000"], check=True, capture_output=True, text=True).stdout
    assert "PASS (0 mismatches)" in out, out
    ties = int(out.split("ties met by the serial loops:")[1].split()[0])
    assert ties > 10000, out                       # the tie path is really exercised
