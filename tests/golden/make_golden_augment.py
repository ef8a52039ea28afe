"""Mint tests/golden/augment.pt from the REFERENCE's augmentation helpers (delta_experiment/scripts/common.py:1161-1314), run in
the build container where /root/reference exists; the fixture is data only (inputs, index lists, names, scalars).  Rotated
pixels are NOT in it: the reference's rotation is torchvision's, which is not installed here (tta/augment.py restates it)."""
import sys
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
from make_golden import REF, _stub_longcat  # noqa: E402


def main():
    _stub_longcat()
    sys.path.insert(0, str(REF / "delta_experiment" / "scripts"))
    import common as C  # noqa
    out = {}
    out["parse_speed_factors"] = [[s, C.parse_speed_factors(s)] for s in ("", "0.5,2.0", " 2 , ,0.25", "3", "1.0,0.5")]
    out["rotation_scale"] = [[h, w, d, C._rotation_scale(h, w, d)] for (h, w) in ((480, 832), (720, 1280), (64, 64), (9, 31))
                             for d in (-15.0, -7.5, 0.0, 3.0, 10.0, 45.0)]
    g = torch.Generator().manual_seed(11)
    clip = torch.rand(1, 3, 13, 6, 10, generator=g) * 2 - 1
    cases = []
    for kw in (dict(enable_flip=True, rotate_deg=0.0, rotate_random_count=0, speed_factors=[0.5, 2.0, 1.0, 3.4, 0.26]),
               dict(enable_flip=False, rotate_deg=0.0, rotate_random_count=0, speed_factors=None),
               dict(enable_flip=True, rotate_deg=0.0, rotate_random_count=0, speed_factors=[2.6])):
        vs = C.build_augmented_pixel_variants(clip, **kw)
        cases.append({"kw": kw, "names": [v["name"] for v in vs], "frames": [v["pixel_frames"].clone() for v in vs]})
    out["clip"] = clip
    out["pixel_variants"] = cases
    # the random-angle draw (no rotation applied: only what is drawn and how the variants would be named)
    draws = []
    for seed, (rmin, rmax, cnt, step) in enumerate(((5.0, 15.0, 2, 1.0), (15.0, 5.0, 4, 2.5), (0.0, 0.0, 2, 1.0), (3.0, 9.0, 3, 0.0))):
        torch.manual_seed(100 + seed)
        lo, hi = (rmin, rmax) if rmin <= rmax else (rmax, rmin)
        if step and step > 0:
            options = torch.arange(lo, hi + 1e-6, step)
            if len(options) == 0:
                options = torch.tensor([lo])
            idx = torch.randint(0, len(options), (cnt,))
            angles = options[idx].tolist()
        else:
            angles = torch.empty(cnt).uniform_(lo, hi).tolist()
        draws.append({"seed": 100 + seed, "args": [rmin, rmax, cnt, step], "angles": angles})
    out["angle_draws"] = draws          # the same statements as common.py:1280-1290, recorded with their seeds
    torch.save(out, HERE / "augment.pt")
    print("wrote augment.pt:", [c["names"] for c in cases])


if __name__ == "__main__":
    main()
