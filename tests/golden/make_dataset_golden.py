"""Build container only: which videos the REFERENCE's dataset listers select, and in which order, on small trees of empty files
(`validate_decodable=False`: no file is opened) -> tests/golden/dataset_selection.json.  The trees are described in the fixture by
their relative file names and metadata rows, so the test can rebuild them anywhere."""
import json
import sys
import tempfile
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import _ref_loader as RL  # noqa: E402

RL.stub_longcat()
RL.add_reference_paths()
import common as ref  # noqa: E402  (delta_experiment/scripts/common.py)


def tree_ucf(n_classes, per_class, ext=".avi"):
    return [f"UCF-101/Class_{c:02d}/v_Class_{c:02d}_g{(i // 3) + 1:02d}_c{(i % 3) + 1:02d}{ext}" for c in range(n_classes) for i in range(per_class[c % len(per_class)])]


CASES = [
    {"name": "ucf_scan_stratified", "root": "data_ucf", "files": tree_ucf(7, [5, 3, 9]), "meta": None, "fn": "ucf", "kw": {"max_videos": 12, "seed": 42}},
    {"name": "ucf_scan_stratified_topup", "root": "data_ucf", "files": tree_ucf(4, [2, 9]), "meta": None, "fn": "ucf", "kw": {"max_videos": 15, "seed": 7}},
    {"name": "ucf_scan_plain", "root": "data_ucf", "files": tree_ucf(5, [4]), "meta": None, "fn": "ucf", "kw": {"max_videos": 6, "seed": 3, "stratified": False}},
    {"name": "ucf_more_classes_than_videos", "root": "data_ucf", "files": tree_ucf(9, [2]), "meta": None, "fn": "ucf", "kw": {"max_videos": 4, "seed": 11}},
    {"name": "singleton_classes_fall_back", "root": "data_misc", "files": [f"c{i:02d}/clip{i:02d}.mp4" for i in range(14)], "meta": None, "fn": "ucf",
     "kw": {"max_videos": 10, "seed": 5}},
    {"name": "mp4_then_avi_and_root_level", "root": "data_mix", "files": ["z.avi", "a.mp4", "k/b.avi", "k/c.mp4", "m/d.mp4"], "meta": None, "fn": "ucf",
     "kw": {"max_videos": 5, "seed": 1}},
    {"name": "panda_path_disables_stratified", "root": "Panda70M_subset", "files": [f"videos/p{i:03d}.mp4" for i in range(20)],
     "meta": {"header": ["filename", "caption", "category"],
              "rows": [[f"p{i:03d}.mp4", ("['a cat', 'b']" if i % 3 == 0 else f"caption {i}"), f"cat{i % 4}"] for i in range(20)]},
     "fn": "ucf", "kw": {"max_videos": 8, "seed": 42}},
    {"name": "metadata_text_and_class_name_columns_missing_files", "root": "data_meta", "files": [f"videos/m{i:02d}.mp4" for i in range(10)] + ["n00.mp4"],
     "meta": {"header": ["video_path", "text", "class_name"],
              "rows": [[f"m{i:02d}.mp4", f"  t{i} ", "x" if i < 6 else "y"] for i in range(10)] + [["n00.mp4", "[]", "y"], ["gone.mp4", "nope", "x"]]},
     "fn": "ucf", "kw": {"max_videos": 7, "seed": 9}},
    {"name": "panda_lister_scan", "root": "panda_scan", "files": [f"s/{i:02d}.mp4" for i in range(9)] + ["s/skip.avi"], "meta": None, "fn": "panda",
     "kw": {"max_videos": 5, "seed": 2}},
]


def build(root: Path, case):
    for rel in case["files"]:
        p = root / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_bytes(b"")
    if case["meta"]:
        import csv
        with open(root / "metadata.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(case["meta"]["header"])
            w.writerows(case["meta"]["rows"])


def main():
    out = []
    for case in CASES:
        with tempfile.TemporaryDirectory() as td:
            root = Path(td) / case["root"]
            root.mkdir()
            build(root, case)
            if case["fn"] == "ucf":
                got = ref.load_ucf101_video_list(str(root), **case["kw"])
            else:
                got = ref.load_panda70m_video_list(str(root), **case["kw"])
            rec = dict(case)
            rec["selected"] = [{"rel": str(Path(e["video_path"]).relative_to(root)), "caption": e["caption"], "class_name": e["class_name"]} for e in got]
            out.append(rec)
            print(case["name"], len(got), [Path(e["video_path"]).name for e in got][:6])
    (HERE / "dataset_selection.json").write_text(json.dumps({"cases": out, "captions": [
        [repr(x), ref._normalize_caption(x)] for x in (None, "", "  a ", ["", " b"], ("c",), "['d', 'e']", "[]", "['', '  ']", "[1, 2]", "[oops", "x]")]}, indent=1))


if __name__ == "__main__":
    main()
