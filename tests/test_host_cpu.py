"""CPU tests of the C++ host layer (host/Evolutionary_Strategy.hpp, Benchmarker.hpp,
CSV_Logger.hpp): compiled with g++ here, no GPU and no libsots_hip involved, compared with the
CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd", "host")


@pytest.fixture(scope="module")
def run(tmp_path_factory):
    d = tmp_path_factory.mktemp("hostcpu")
    exe = d / "host_cpu_test"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wextra", "-Wno-unused-parameter",
                           "-o", str(exe), os.path.join(HOST, "host_cpu_test.cpp")])
    out = subprocess.run([str(exe), str(d)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    return d, out.stdout


def f32(d, name):
    return np.fromfile(d / name, np.float32)


def test_objective_matches_the_oracle(run, O):
    d, text = run
    assert np.array_equal(f32(d, "wavetable.f32"), O.wavetable())
    w64, wf = O.window(1024)
    assert np.array_equal(f32(d, "window.f32"), w64.astype(np.float32))
    p2 = [3520.0, 8.0, 3520.0, 1.0]
    a = O.synth(0, [0.411931818, 0.375, 0.0568181818, 1.0], [0.0] * 4, p2, 1024)
    assert np.array_equal(f32(d, "voice2.f32"), a)
    # fp64 FFTs of different construction: equal to the last fp32 bit of the peak
    np.testing.assert_allclose(f32(d, "voice2_mag.f32"), O.spectrum(a), rtol=0, atol=np.spacing(np.float32(0.25)))
    # phases restart on every call (the reference carries them over, Evolutionary_Strategy.hpp:178-180)
    assert np.array_equal(f32(d, "voice2b.f32"), O.synth(0, [0.9, 1.0, 0.02, 0.5], [0.0] * 4, p2, 1024))
    assert np.array_equal(f32(d, "voice6.f32"),
                          O.synth(1, [3078 / 3520, 2 / 8, 3015 / 3520, 1.5 / 8, 3141 / 3520, 1 / 8], [0.0] * 6,
                                  [3520.0, 8.0] * 3, 1024))
    assert np.array_equal(f32(d, "voice8.f32"),
                          O.synth(3, [0.3, 0.25, 0.85, 0.19, 0.89, 0.125, 0.5, 0.1], [0.0] * 8, [3520.0, 8.0] * 4, 1024))
    assert np.array_equal(f32(d, "voice12.f32"),
                          O.synth(2, [0.41, 0.375, 0.057, 1.0, 0.2, 0.5, 0.11, 0.7, 0.6, 0.1, 0.3, 0.4], [0.0] * 4, p2, 1024))
    line = [l for l in text.splitlines() if l.startswith("scale")][0].split()
    assert [float(x) for x in line[1:5]] == [1760.0, 2.0, 3520.0, 0.0]
    assert float(line[6]) == 1.0 and int(line[8]) == 1024 + 8


def test_population_sort_is_stable_and_constants(run):
    _, text = run
    line = [l for l in text.splitlines() if l.startswith("order")][0].split()
    # fitness 3,1,NaN,1,0,-0,2,1 -> 0 and -0 tie (index order), the three 1s keep their order, NaN last
    assert [int(float(x)) for x in line[1:9]] == [4, 5, 1, 3, 7, 6, 0, 2]
    alpha, inv, rtop, bscale, beta = [float(x) for x in line[10:15]]
    assert alpha == np.float32(1.4) and inv == np.float32(1.0) / np.float32(1.4)
    assert rtop == np.sqrt(np.float32(2.0) / np.float32(np.pi)) and bscale == 0.5 and beta == np.sqrt(np.float32(0.5))


def test_benchmarker_and_csv(run):
    d, text = run
    assert "counts 3 0 total 9.000" in text and "after 0" in text and "reject 0 accept 1" in text
    rows = (d / "bench.csv").read_text().strip().splitlines()
    assert rows[0] == "Test_Name,Total_Time,Average_Time,Max_Time,Min_Time,Max_Difference,Average_Difference,"
    a = rows[1].rstrip(",").split(",")
    assert a[0] == "stageA" and [float(x) for x in a[1:5]] == [9.0, 3.0, 4.0, 2.0]
    # differences between consecutive samples: |2-0|, |4-2|, |3-4| -> max 2, mean 5/3
    assert float(a[5]) == 2.0 and abs(float(a[6]) - 5.0 / 3.0) < 1e-6
    once = rows[2].rstrip(",").split(",")
    assert once[0] == "once" and len(once) == 7 and float(once[2]) == 7.5
    assert rows[3].startswith("wall,")
    assert (d / "raw.csv").read_text() == "a,b,\n1,2,\n"


def test_benchmarker_rows_equal_the_reference_class(tmp_path):
    """The one piece of the boundary the reference itself can pin: the golden CSVs are what
    /root/reference/Benchmarker.hpp + CSV_Logger.hpp - compiled as they stand by tests/golden/make_benchmarker_golden.sh -
    write for tests/golden/benchmarker_sequence.inc; host/Benchmarker.hpp is fed the same sequence here.
      * benchmarker_rows_v1_absdouble.csv (the reference with the double overload of abs in scope): every row, all seven
        fields, byte for byte - including what carries over an elapsedTimer reset (Benchmarker.hpp:151,164-166);
      * benchmarker_rows_v1.csv (plain g++: `abs` resolves to int abs(int), Benchmarker.hpp:66,104,127): the five fields in
        front byte for byte; its two difference columns are the truncated-to-whole-ms version of the other file's and are
        not reproduced;
      * the one deliberate difference: the timer that fired once is a 6-field record the reference's CSV_Logger drops
        (Benchmarker.hpp:145-157, CSV_Logger.hpp:30-31); here it is a full row."""
    gold = os.path.join(ROOT, "tests", "golden")
    exe = tmp_path / "benchmarker_driver"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", HOST, "-I", gold, "-o", str(exe),
                           os.path.join(gold, "benchmarker_driver.cpp")])
    out = tmp_path / "rows.csv"
    subprocess.run([str(exe), str(out)], check=True, capture_output=True, timeout=60)
    mine = out.read_text().splitlines()
    once = [r for r in mine if r.startswith("once,")]
    assert len(once) == 1 and once[0] == "once,7.500000,7.500000,7.500000,7.500000,7.500000,7.500000,"
    mine = [r for r in mine if not r.startswith("once,")]
    absdouble = open(os.path.join(gold, "benchmarker_rows_v1_absdouble.csv")).read().splitlines()
    plain = open(os.path.join(gold, "benchmarker_rows_v1.csv")).read().splitlines()
    assert len(absdouble) == len(plain) == 7 and not any(r.startswith("once,") for r in absdouble + plain)
    assert mine == absdouble
    for a, b in zip(mine, plain):
        assert a.split(",")[:5] == b.split(",")[:5]
    # the plain build's difference columns never exceed the double build's (every difference was truncated towards zero)
    for a, b in zip(absdouble[1:], plain[1:]):
        assert float(b.split(",")[5]) <= float(a.split(",")[5])


def test_wav_writer_and_reader_equal_the_reference_audiofile(tmp_path):
    """SURVEY 8(f)2 pinned by the reference itself: tests/golden/audiofile_24bit_v1.wav is what /root/reference/AudioFile.cpp -
    compiled as it stands by tests/golden/make_wav_golden.sh - writes for tests/golden/wav_samples.inc through the calls of the
    reference's outputAudioFile (main.cpp:337-366).  host/Wav_IO.hpp must write the same bytes (header and truncated 24-bit
    samples, AudioFile.cpp:595) and read the golden file back to what AudioFile::load returns (integer / 8388608, :349-358).
    The one deliberate difference is outside the fixture: a sample of +1.0 or beyond wraps around in the reference (no clamp); here
    it is clamped to 0x7FFFFF / 0x800000."""
    gold = os.path.join(ROOT, "tests", "golden")
    exe = tmp_path / "wav_driver"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", HOST, "-I", gold, "-o", str(exe), os.path.join(gold, "wav_driver.cpp")])
    out = tmp_path / "mine.wav"
    res = subprocess.run([str(exe), str(out), os.path.join(gold, "audiofile_24bit_v1.wav")], check=True, capture_output=True, text=True, timeout=60)
    assert "read back 1536 samples, 0 differ" in res.stdout
    mine, ref = out.read_bytes(), open(os.path.join(gold, "audiofile_24bit_v1.wav"), "rb").read()
    assert len(ref) == 44 + 3 * 1536 and mine == ref
    # the clamp (the reference would write 0x800000 = -1.0 for +1.0): a two-sample file through a tiny driver
    src = tmp_path / "clamp.cpp"
    src.write_text('#include "Wav_IO.hpp"\nint main(int, char **a) { const float x[4] = {1.0f, 7.0f, -3.0f, -1.0f}; outputAudioFile(a[1], x, 4); return 0; }\n')
    exe2 = tmp_path / "clamp"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", HOST, "-o", str(exe2), str(src)])
    subprocess.run([str(exe2), str(tmp_path / "c.wav")], check=True, timeout=60)
    data = (tmp_path / "c.wav").read_bytes()[44:]
    assert data == bytes([0xFF, 0xFF, 0x7F] * 2 + [0x00, 0x00, 0x80] * 2)
