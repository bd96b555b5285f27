"""bench.py prints ONE stdout line that the driver parses: it must stay short (round 5's 21 KB line was not parsed at all) and
carry the contract's keys, whatever the side sections measured.  The line is built here from a recorded run."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _recorded():
    with open(os.path.join(ROOT, "profiles", "r5_bench_n1.json")) as f:
        return json.load(f)


def test_compact_line_is_short_and_round_trips():
    out = _recorded()
    assert len(json.dumps(out)) > 16000                       # the record that broke the driver's parser
    line = json.dumps(bench.compact_line(out))
    assert len(line) < 4096, len(line)
    back = json.loads(line)
    for key in bench.DRIVER_KEYS:
        assert key in back, key
    assert back["value"] > 0 and back["unit"] == "samples/s" and back["higher_is_better"] is True and back["vs_baseline"] is None
    assert set(back["config"]) >= {"workload"} and "model" not in back["config"]
    rf = back["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and "traffic" in rf
    cb = back["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] == 1 and cb["value"] > 0 and cb["sample"]
    assert len(back["paths"]) >= 10
    for name, row in back["paths"].items():
        assert "note" not in row and all(not isinstance(v, (list, dict)) for v in row.values()), name
        assert "value" in row and "unit" in row, name


def test_emit_writes_the_full_record_aside(tmp_path, monkeypatch, capsys):
    out = _recorded()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.emit(out)
    cap = capsys.readouterr()
    lines = [l for l in cap.out.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096
    assert json.loads(lines[0])["extra_file"] == bench.EXTRA_FILE
    with open(tmp_path / bench.EXTRA_FILE) as f:
        assert json.load(f) == out                              # nothing measured is lost: it moved
    assert "[bench extra] " in cap.err


def test_a_huge_side_section_cannot_lengthen_the_line():
    out = _recorded()
    out["extra"]["itemknn"]["emulated_8_way"]["kernel_ms_per_piece"] = [[0.123456789] * 64 for _ in range(64)]
    out["extra"]["paths"]["slim_bpr_dense"]["note"] = "x" * 50000
    assert len(json.dumps(bench.compact_line(out))) < 4096


def test_numbers_carry_five_significant_digits():
    assert bench._sig(213053896.0011929) == 213050000.0 and bench._sig(0.08492590706) == 0.084926
    assert bench._sig(float("nan")) is None and bench._sig(7) == 7 and bench._sig(True) is True and bench._sig("s") == "s"
