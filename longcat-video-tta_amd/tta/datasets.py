"""Which videos a run evaluates, in which order: the index logic of the reference's dataset listers (SURVEY Appendix A lists the
shuffle among the integer results that must be bit-exact — two runs over the same `--data-dir --max-videos --seed` must adapt the
SAME clips in the SAME order or their summaries are not comparable).

Mirrors `load_ucf101_video_list` (delta_experiment/scripts/common.py:814-947; every runner calls it, also on Panda-70M trees:
run_lora_tta.py:911-913) and `load_panda70m_video_list` (:950-1011): entries from `metadata.csv` (columns filename | video_path,
caption | text, category | class_name) or, without one, from a recursive scan for *.mp4 then *.avi; one `numpy.random.RandomState(seed)`
drives every shuffle; stratified sampling takes max_videos // n_classes per class in sorted class order, tops up from the shuffled
leftovers and shuffles the selection; it is switched off for a path containing "panda" and when most classes are singletons.
`validate_decodable` (PyAV opens every file and decodes one frame) needs `av`; where it is not installed the listing is returned
unvalidated and says so.  Host-side only: nothing here touches the GPU."""
import ast
import csv
from collections import Counter, defaultdict
from pathlib import Path
from typing import Any, Dict, List, Optional

import numpy as np


def normalize_caption(raw: Any) -> str:
    """common.py:783-812: a caption cell may be a string, a list, or the repr of a list; the first non-empty item wins."""
    def first(items):
        for it in items:
            t = str(it).strip()
            if t:
                return t
        return ""
    if raw is None:
        return ""
    if isinstance(raw, (list, tuple)):
        return first(raw)
    s = str(raw).strip()
    if s.startswith("[") and s.endswith("]"):
        try:
            parsed = ast.literal_eval(s)
        except (ValueError, SyntaxError):
            return s
        if isinstance(parsed, (list, tuple)):
            got = first(parsed)
            return got if got else s
    return s


def _resolve(data_dir: Path, fname: str) -> Optional[Path]:
    for cand in (data_dir / "videos" / fname, data_dir / fname):
        if cand.exists():
            return cand
    return None


def _rows_of_metadata(root: Path, meta: Path, class_of, errors=None) -> List[Dict]:
    """Entries of a metadata file whose video exists (under `videos/` or beside the file); [] when there is no such file."""
    if not meta.exists():
        return []
    kw = {"encoding": "utf-8", "errors": errors} if errors else {}
    out = []
    with open(meta, "r", **kw) as f:
        for row in csv.DictReader(f):
            vp = _resolve(root, row.get("filename", row.get("video_path", "")))
            if vp is not None:
                out.append({"video_path": str(vp), "caption": normalize_caption(row.get("caption", row.get("text", ""))),
                            "class_name": class_of(row)})
    return out


def _keep_decodable(entries: List[Dict], what: str) -> List[Dict]:
    try:
        import av
    except ImportError:
        print(f"  (PyAV is not installed: {what} listing not validated for decodability)")
        return entries
    good, bad = [], []
    for e in entries:
        ok = True
        try:
            c = av.open(str(e["video_path"]))
            try:
                next(c.decode(video=0))
            except StopIteration:
                ok = False
            finally:
                c.close()
        except Exception:
            ok = False
        (good if ok else bad).append(e)
    if bad:
        print(f"  Dropped {len(bad)} undecodable videos during dataset load.")
        for e in bad[:5]:
            print(f"    bad_video: {e['video_path']}")
    return good


def load_ucf101_video_list(data_dir: str, max_videos: int = 100, seed: int = 42, stratified: bool = True,
                           validate_decodable: bool = False) -> List[Dict]:
    root = Path(data_dir)
    meta = root / "metadata.csv"
    entries = _rows_of_metadata(root, meta, lambda row: row.get("category", row.get("class_name", "unknown")), errors="replace")
    if entries:
        print(f"  Loaded {len(entries)} videos from {meta}")
    if not entries:
        for pattern in ("*.mp4", "*.avi"):
            for vp in sorted(root.rglob(pattern)):
                cls = vp.parent.name if vp.parent != root else "unknown"
                entries.append({"video_path": str(vp), "caption": cls.replace("_", " "), "class_name": cls})
    if not entries:
        raise FileNotFoundError(f"No video files found in {root}")
    if validate_decodable:
        entries = _keep_decodable(entries, "dataset")
        if not entries:
            raise FileNotFoundError(f"No decodable video files found in {root}")
    rng = np.random.RandomState(seed)
    if stratified and "panda" in str(root).lower():
        print("  Stratified sampling disabled for Panda dataset path.")
        stratified = False
    if not stratified:
        rng.shuffle(entries)
        return entries[:max_videos]
    by_class = defaultdict(list)
    for e in entries:
        by_class[e["class_name"]].append(e)
    n_classes = len(by_class)
    sizes = Counter(e["class_name"] for e in entries)
    singleton_ratio = sum(1 for c in sizes.values() if c == 1) / max(n_classes, 1)
    if singleton_ratio > 0.5 and n_classes > max_videos // 2:
        print(f"  Stratified sampling disabled: many singleton classes (classes={n_classes}, singleton_ratio={singleton_ratio:.2f}).")
        rng.shuffle(entries)
        return entries[:max_videos]
    per_class = max(1, max_videos // n_classes)
    selected, leftover = [], []
    for cls in sorted(by_class):
        members = by_class[cls]
        rng.shuffle(members)
        selected.extend(members[:per_class])
        leftover.extend(members[per_class:])
    if len(selected) < max_videos and leftover:
        rng.shuffle(leftover)
        selected.extend(leftover[: max_videos - len(selected)])
    rng.shuffle(selected)
    return selected[:max_videos]


def load_panda70m_video_list(data_dir: str, meta_path: Optional[str] = None, max_videos: int = 100, seed: int = 42,
                             validate_decodable: bool = False) -> List[Dict]:
    root = Path(data_dir)
    if meta_path and Path(meta_path).exists():
        entries = _rows_of_metadata(root, Path(meta_path), lambda row: "panda70m")
    else:
        entries = [{"video_path": str(vp), "caption": "A video clip", "class_name": "panda70m"} for vp in sorted(root.rglob("*.mp4"))]
    if validate_decodable:
        entries = _keep_decodable(entries, "Panda")
    rng = np.random.RandomState(seed)
    rng.shuffle(entries)
    return entries[:max_videos]


# ---- caption quality guard and caption override (common.py:1019-1157; every runner calls both right after the listing:
# run_lora_tta.py:914-925).  A listing whose captions are mostly empty, mostly identical or dominated by a placeholder is a
# mis-joined metadata file, and adapting 100 videos to "a video clip" is an expensive way to find out.
GENERIC_CAPTIONS = {"", "video", "videos", "a video", "a video clip", "video clip", "unknown", "none", "nan"}


def analyze_caption_quality(video_entries: List[Dict[str, Any]], top_k: int = 5) -> Dict[str, Any]:
    total = len(video_entries)
    nonempty = [c for c in (str(v.get("caption", "")).strip() for v in video_entries) if c]
    counts = Counter(c.lower() for c in nonempty)
    n = len(nonempty)
    top = counts.most_common(max(int(top_k), 1))
    top1_caption, top1_count = top[0] if top else ("", 0)
    return {"total": total, "nonempty_count": n, "nonempty_ratio": (n / total) if total else 0.0, "unique_count": len(counts),
            "unique_ratio": (len(counts) / n) if n else 0.0, "top1_caption": top1_caption, "top1_count": top1_count,
            "top1_ratio": (top1_count / n) if n else 0.0, "avg_caption_len": float(np.mean([len(c) for c in nonempty])) if nonempty else 0.0,
            "top_captions": top}


def validate_caption_quality(video_entries: List[Dict[str, Any]], *, mode: str = "fail", min_nonempty_ratio: float = 0.95,
                             min_unique_ratio: float = 0.10, max_top1_ratio: float = 0.50, max_generic_top1_ratio: float = 0.20,
                             top_k: int = 5, context: str = "") -> Dict[str, Any]:
    mode = (mode or "fail").lower()
    if mode not in {"fail", "warn", "off"}:
        raise ValueError(f"Invalid caption guard mode: {mode}")
    st = analyze_caption_quality(video_entries, top_k=top_k)
    prefix = f"[caption_guard:{context}]" if context else "[caption_guard]"
    print(f"{prefix} total={st['total']} nonempty_ratio={st['nonempty_ratio']:.4f} unique_ratio={st['unique_ratio']:.4f} "
          f"top1_ratio={st['top1_ratio']:.4f} avg_len={st['avg_caption_len']:.1f}")
    if st["top_captions"]:
        print(f"{prefix} top captions:")
        for cap, count in st["top_captions"]:
            print(f"  - {count:4d} | {cap[:180]}")
    if mode == "off" or st["total"] < 20:          # small listings are never judged
        return st
    floors = (("nonempty_ratio", float(min_nonempty_ratio)), ("unique_ratio", float(min_unique_ratio)))
    reasons = [f"{k}={st[k]:.4f} < {lim:.4f}" for k, lim in floors if st[k] < lim]
    if st["top1_ratio"] > float(max_top1_ratio):
        reasons.append(f"top1_ratio={st['top1_ratio']:.4f} > {float(max_top1_ratio):.4f}")
    if st["top1_caption"] in GENERIC_CAPTIONS and st["top1_ratio"] > float(max_generic_top1_ratio):
        reasons.append(f"generic top caption dominates ('{st['top1_caption']}' ratio={st['top1_ratio']:.4f} > "
                       f"{float(max_generic_top1_ratio):.4f})")
    if reasons:
        msg = f"{prefix} suspicious captions detected: " + "; ".join(reasons)
        if mode == "warn":
            print(f"WARNING: {msg}")
        else:
            raise RuntimeError(msg)
    return st


def apply_fixed_caption(video_entries: List[Dict[str, Any]], fixed_caption: Optional[str], *, context: str = "eval") -> List[Dict[str, Any]]:
    if fixed_caption is None:
        return video_entries
    cap = str(fixed_caption).strip()
    if len(cap) >= 2 and cap[0] == cap[-1] and cap[0] in ("'", '"'):      # a shell-quoted literal that arrived with its quotes
        cap = cap[1:-1]
    for row in video_entries:
        row["caption"] = cap
    print(f"[caption_override:{context}] applied fixed caption to {len(video_entries)} videos: {cap!r}")
    return video_entries
