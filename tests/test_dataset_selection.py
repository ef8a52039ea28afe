"""CPU: the drop-in selects the SAME videos in the SAME order as the reference's dataset listers (the shuffle is
on the list of integer results that must be bit-exact).  tests/golden/dataset_selection.json holds what the reference's
`load_ucf101_video_list` / `load_panda70m_video_list` returned on small trees of empty files (make_dataset_golden.py); the trees
are rebuilt here from the fixture's own file lists."""
import csv
import json
import sys
import tempfile
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "longcat-video-tta_amd"))
from tta import datasets as DS  # noqa: E402

FIX = json.loads((ROOT / "tests" / "golden" / "dataset_selection.json").read_text())


@pytest.mark.parametrize("case", FIX["cases"], ids=[c["name"] for c in FIX["cases"]])
def test_selection_and_order_equal_the_reference(case):
    with tempfile.TemporaryDirectory() as td:          # (not pytest's tmp_path: its name carries the test id, and a path that
        root = Path(td) / case["root"]                  #  contains "panda" switches stratified sampling off)
        root.mkdir()
        for rel in case["files"]:
            p = root / rel
            p.parent.mkdir(parents=True, exist_ok=True)
            p.write_bytes(b"")
        if case["meta"]:
            with open(root / "metadata.csv", "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(case["meta"]["header"])
                w.writerows(case["meta"]["rows"])
        fn = DS.load_ucf101_video_list if case["fn"] == "ucf" else DS.load_panda70m_video_list
        got = fn(str(root), **case["kw"])
        got = [{"rel": str(Path(e["video_path"]).relative_to(root)), "caption": e["caption"], "class_name": e["class_name"]} for e in got]
    assert got == case["selected"]


def test_caption_normalisation_equals_the_reference():
    for src, want in FIX["captions"]:
        assert DS.normalize_caption(eval(src)) == want, src


def test_an_empty_tree_raises_like_the_reference():
    with tempfile.TemporaryDirectory() as td:
        with pytest.raises(FileNotFoundError, match="No video files found"):
            DS.load_ucf101_video_list(td)


def test_runner_listing_follows_the_sampler_and_maps_to_pre_encoded_clips():
    """`list_eval_entries` (every runner's first step): the reference sampler picks and orders, each pick is mapped to its pre-encoded
    `<data-dir>/latents/<stem>.pt`; picks without one are reported and skipped; a latents-only directory keeps its sorted order."""
    import types
    from tta import runner_common as R
    case = next(c for c in FIX["cases"] if c["name"] == "ucf_scan_stratified")
    with tempfile.TemporaryDirectory() as td:
        root = Path(td) / "data_ucf"
        (root / "latents").mkdir(parents=True)
        for rel in case["files"]:
            p = root / rel
            p.parent.mkdir(parents=True, exist_ok=True)
            p.write_bytes(b"")
        want = [Path(e["rel"]).stem for e in case["selected"]]
        for stem in want[:-2]:                                    # the last two picks were never pre-encoded
            (root / "latents" / f"{stem}.pt").write_bytes(b"")
        args = types.SimpleNamespace(data_dir=str(root), max_videos=case["kw"]["max_videos"], seed=case["kw"]["seed"])
        got = R.list_eval_entries(args, None)
        assert [e["name"] for e in got] == want[:-2] and all(e["kind"] == "latents" for e in got)
        assert got[0]["class_name"] == case["selected"][0]["class_name"]
        only = Path(td) / "only_latents"
        (only / "latents").mkdir(parents=True)
        for n in ("b", "a", "c"):
            (only / "latents" / f"{n}.pt").write_bytes(b"")
        args2 = types.SimpleNamespace(data_dir=str(only), max_videos=2, seed=42)
        assert [e["name"] for e in R.list_eval_entries(args2, None)] == ["a", "b"]


CAP = json.loads((ROOT / "tests" / "golden" / "caption_guard.json").read_text())


@pytest.mark.parametrize("i", range(len(CAP["cases"])))
def test_caption_guard_equals_the_reference(i):
    """Statistics, thresholds and the raised message of `validate_caption_quality` (common.py:1035-1137) on seeded caption lists."""
    c = CAP["cases"][i]
    entries = [dict(e) for e in CAP["lists"][c["list"]]]
    if "raises" in c:
        with pytest.raises(RuntimeError) as ei:
            DS.validate_caption_quality(entries, **c["kw"])
        assert str(ei.value) == c["raises"]
    else:
        st = DS.validate_caption_quality(entries, **c["kw"])
        want = dict(c["stats"])
        assert [list(t) for t in st.pop("top_captions")] == want.pop("top_captions")
        assert st == want


def test_fixed_caption_override_and_generic_list_equal_the_reference():
    for c in CAP["fixed"]:
        rows = [{"caption": "one"}, {"caption": "two", "k": 1}]
        assert [r["caption"] for r in DS.apply_fixed_caption(rows, c["fixed_caption"])] == c["captions"]
    assert sorted(DS.GENERIC_CAPTIONS) == CAP["generic"]
    with pytest.raises(ValueError, match="Invalid caption guard mode"):
        DS.validate_caption_quality([], mode="strict")


def test_runner_listing_applies_the_override_and_runs_the_guard():
    import types
    from tta import runner_common as R
    with tempfile.TemporaryDirectory() as td:
        root = Path(td) / "data"
        (root / "videos").mkdir(parents=True)
        rows = [[f"v{i:02d}.mp4", "A video clip" if i < 20 else f"a person does thing {i}", f"c{i % 3}"] for i in range(30)]
        for r in rows:
            (root / "videos" / r[0]).write_bytes(b"")
        with open(root / "metadata.csv", "w", newline="") as f:
            w = csv.writer(f); w.writerow(["filename", "caption", "category"]); w.writerows(rows)
        (root / "latents").mkdir()
        for r in rows:
            (root / "latents" / (Path(r[0]).stem + ".pt")).write_bytes(b"")
        base = dict(data_dir=str(root), max_videos=30, seed=42)
        with pytest.raises(RuntimeError, match=r"\[caption_guard:eval\] suspicious captions detected: .*generic top caption dominates"):
            R.list_eval_entries(types.SimpleNamespace(**base), None)                         # mode defaults to fail
        got = R.list_eval_entries(types.SimpleNamespace(caption_guard_mode="warn", **base), None)
        assert len(got) == 30
        got = R.list_eval_entries(types.SimpleNamespace(fixed_caption='"a juggler"', caption_guard_mode="off", **base), None)
        assert {e["caption"] for e in got} == {"a juggler"}
