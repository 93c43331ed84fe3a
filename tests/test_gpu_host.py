"""GPU tests of the C++ host side: Evolutionary_Strategy_HIP driven through the reference's
base-class interface (host/host_test.cpp) and the sots_match command line."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd")


def test_evolutionary_strategy_hip_through_base_class(tmp_path):
    exe = os.path.join(PKG_DIR, "sots_host_test")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["sorted"] and r["aos_ok"] and r["fused_equals_staged"] and r["bad_config_throws"]
    assert r["chunks"] == 2
    assert r["csv_header"].startswith("Test_Name,Total_Time,Average_Time,Max_Time,Min_Time,Max_Difference,Average_Difference")
    assert r["csv_has_total"] and r["csv_rows"] >= 4
    # per-launch rows: 2 chunks x 40 generations of the fused loop = 80 sort launches behind the averages
    assert r["sort_total_ms"] > 0 and abs(r["sort_avg_ms"] * 80 - r["sort_total_ms"]) < 0.02 * r["sort_total_ms"]
    # un-instrumented mode (isBenchmarking false): no hipEvents, no stage rows, only the total + a rate
    assert r["quiet_stage_rows"] == 0 and r["quiet_rows"] == 1 and r["quiet_has_total"]
    assert r["quiet_candidates_per_s"] > 0
    # numDevices = 2 (both islands on device 0): 2 x 8192 candidates, 30 generations, elites exchanged every generation
    assert r["group_islands"] == 2 and r["group_tail_sorted"] and r["group_candidates_per_s"] > 0
    assert r["group_best"] < 1e-4
    # 40 generations of 8192 candidates: the CPU oracle reaches 3.3e-9 on chunk 0 and the local
    # optimum 0.0258 on chunk 1 with this seed; the best of a random population is ~0.1-0.3
    assert r["host_fitness_chunk0"] < 1e-6
    assert r["best_fitness_last_chunk"] < 3e-2


def read_wav24(path):
    d = open(path, "rb").read()
    assert d[:4] == b"RIFF" and d[8:12] == b"WAVE"
    fmt, ch, rate, _, _, bits = struct.unpack("<HHIIHH", d[20:36])
    n = struct.unpack("<I", d[40:44])[0] // 3
    raw = np.frombuffer(d[44:44 + 3 * n], np.uint8).reshape(n, 3).astype(np.int32)
    v = (raw[:, 0] | (raw[:, 1] << 8) | (raw[:, 2] << 16))
    v = np.where(v & 0x800000, v - (1 << 24), v)
    return fmt, ch, rate, bits, v / 8388608.0


def test_sots_match_cli(tmp_path, O):
    exe = os.path.join(PKG_DIR, "sots_match")
    assert os.path.exists(exe)
    cfg = json.load(open(os.path.join(PKG_DIR, "parameters.json")))
    cfg["general"]["outputAudioPath"] = str(tmp_path / "out.wav")
    cfg["general"]["isDebug"] = False
    cfg["evolutionary"]["numGenerations"] = 40
    cfg["evolutionary"]["numParents"], cfg["evolutionary"]["numOffspring"] = 2048, 6144
    p = tmp_path / "parameters.json"
    p.write_text(json.dumps(cfg))
    out = subprocess.run([exe, "-j", str(p)], capture_output=True, text=True, timeout=300, cwd=tmp_path)
    assert out.returncode == 0, out.stderr
    assert "Overall best parameters found" in out.stdout
    fit = float(out.stdout.split("Fitness = ")[1].split()[0])
    assert fit < 1e-6
    fmt, ch, rate, bits, gen = read_wav24(tmp_path / "inputGenerated.wav")
    assert (fmt, ch, rate, bits) == (1, 1, 44100, 24) and len(gen) == 1024
    want = O.synth(0, [1450 / 3520, 3 / 8, 200 / 3520, 1.0], [0.0] * 4, [3520.0, 8.0, 3520.0, 1.0], 1024)
    # the reference's quantisation (AudioFile.cpp:595, pinned by tests/test_host_cpu.py): truncation of sample * 2^23
    q = np.clip(np.trunc(want.astype(np.float32) * np.float32(8388608.0)), -8388608, 8388607)
    assert np.array_equal(gen * 8388608.0, q)
    _, _, _, _, rendered = read_wav24(tmp_path / "out.wav")
    assert len(rendered) == 1 << 14 and np.abs(rendered).max() <= 1.0
    # audio input: match the generated file itself, two chunks
    cfg["type"]["input"] = "audio"
    two = np.concatenate([want, want]).astype(np.float32)
    wav = tmp_path / "in.wav"
    with open(wav, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + two.nbytes) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 3, 1, 44100, 44100 * 4, 4, 32))
        f.write(b"data" + struct.pack("<I", two.nbytes) + two.tobytes())
    cfg["type"]["audio"] = str(wav)
    p.write_text(json.dumps(cfg))
    out = subprocess.run([exe, "-j", str(p)], capture_output=True, text=True, timeout=300, cwd=tmp_path)
    assert out.returncode == 0, out.stderr
    assert float(out.stdout.split("Fitness = ")[1].split()[0]) < 3e-2
    # island model from the JSON block: two islands (sharing device 0), elites every generation
    cfg["type"]["input"] = "params"
    cfg["type"]["HIP"].update({"numDevices": 2, "devices": [0, 0], "numElites": 16, "migrationInterval": 1})
    cfg["general"]["isBenchmarking"] = False
    p.write_text(json.dumps(cfg))
    out = subprocess.run([exe, "-j", str(p)], capture_output=True, text=True, timeout=300, cwd=tmp_path)
    assert out.returncode == 0, out.stderr
    assert float(out.stdout.split("Fitness = ")[1].split()[0]) < 1e-6
    rate = float(out.stdout.split("Candidates evaluated per second: ")[1].split()[0])
    assert rate > 0
    cfg["type"]["HIP"].update({"numDevices": 1})
    cfg["type"]["HIP"].pop("devices")
    # the shortest and a long analysis block (audioLengthLog2 8 and 14, round 4): the driver, the C++ class and the host tables take them
    for log2n in (8, 14):
        cfg["type"]["HIP"].update({"numDevices": 1})
        cfg["type"]["HIP"].pop("devices", None)
        cfg["audio"]["audioLengthLog2"] = log2n
        cfg["evolutionary"]["numGenerations"] = 12
        cfg["evolutionary"]["numParents"], cfg["evolutionary"]["numOffspring"] = 256, 768
        p.write_text(json.dumps(cfg))
        out = subprocess.run([exe, "-j", str(p)], capture_output=True, text=True, timeout=300, cwd=tmp_path)
        assert out.returncode == 0, out.stderr
        assert np.isfinite(float(out.stdout.split("Fitness = ")[1].split()[0]))
        _, _, _, _, gen = read_wav24(tmp_path / "inputGenerated.wav")
        assert len(gen) == 1 << log2n
    cfg["audio"]["audioLengthLog2"] = 10
    # the reference's OpenCL arithmetic from the JSON block (type.HIP.deviceKernelArithmetic): the search runs on it - against a
    # target the host synthesised with the CPU path's arithmetic, so the match is close, not exact
    cfg["type"]["HIP"]["deviceKernelArithmetic"] = True
    cfg["evolutionary"]["numGenerations"] = 40
    cfg["evolutionary"]["numParents"], cfg["evolutionary"]["numOffspring"] = 2048, 6144
    p.write_text(json.dumps(cfg))
    out = subprocess.run([exe, "-j", str(p)], capture_output=True, text=True, timeout=300, cwd=tmp_path)
    assert out.returncode == 0, out.stderr
    fit_dev = float(out.stdout.split("Fitness = ")[1].split()[0])
    assert 1e-9 < fit_dev < 1e-2
    cfg["type"]["HIP"].pop("deviceKernelArithmetic")
    # wrong implementation is refused
    cfg["type"]["implementation"] = "OpenCL"
    p.write_text(json.dumps(cfg))
    out = subprocess.run([exe, "-j", str(p)], capture_output=True, text=True, timeout=60, cwd=tmp_path)
    assert out.returncode != 0 and "HIP backend only" in out.stderr
