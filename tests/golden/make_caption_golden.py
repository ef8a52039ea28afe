"""Build container only: the REFERENCE's caption guard and caption override (delta_experiment/scripts/common.py:1035-1157) on
seeded caption lists -> tests/golden/caption_guard.json (inputs, returned stats, the message of a raised guard)."""
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import _ref_loader as RL  # noqa: E402

RL.stub_longcat()
RL.add_reference_paths()
import common as ref  # noqa: E402

rs = np.random.RandomState(3)
words = ["a", "man", "dog", "runs", "Plays", "guitar", "on", "the", "beach", "video", "clip", "slowly"]
def varied(n): return [" ".join(rs.choice(words, size=rs.randint(2, 7))) for _ in range(n)]
LISTS = {
    "healthy_40": varied(40),
    "small_listing_never_judged": ["x"] * 10,
    "many_empty": varied(20) + [""] * 10 + ["  "] * 3,
    "few_unique": ["Same Caption", "same caption ", "other"] * 12,
    "one_dominates": ["a cat sleeps"] * 30 + varied(20),
    "generic_dominates": ["A video clip"] * 9 + varied(26),
    "mixed_case_and_missing_key": [c.upper() if i % 2 else c for i, c in enumerate(varied(24))],
}
KWS = [{}, {"mode": "warn"}, {"mode": "off"}, {"mode": "fail", "max_top1_ratio": 0.9, "min_unique_ratio": 0.01, "context": "eval"},
       {"mode": "fail", "min_nonempty_ratio": 0.5, "top_k": 2, "context": "retrieval_pool"}]
cases = []
for name, caps in LISTS.items():
    entries = [{"caption": c} for c in caps]
    if name == "mixed_case_and_missing_key":
        entries[3] = {}
    for kw in KWS:
        try:
            st = ref.validate_caption_quality([dict(e) for e in entries], **kw)
            res = {"stats": st}
        except RuntimeError as e:
            res = {"raises": str(e)}
        cases.append({"list": name, "kw": kw, **res})
fixed = []
for fc in (None, "  a person dancing ", '"videos"', "'x'", '"', "it's"):
    rows = [{"caption": "one"}, {"caption": "two", "k": 1}]
    out = ref.apply_fixed_caption(rows, fc, context="eval")
    fixed.append({"fixed_caption": fc, "captions": [r["caption"] for r in out]})
json.dump({"lists": {k: [({"caption": c} if not (k == "mixed_case_and_missing_key" and i == 3) else {}) for i, c in enumerate(v)] for k, v in LISTS.items()},
           "cases": cases, "fixed": fixed, "generic": sorted(ref._GENERIC_CAPTIONS)}, open(HERE / "caption_guard.json", "w"), indent=1)
print(len(cases), "cases;", sum("raises" in c for c in cases), "raise")
