"""A stand-in for PyAV (not installed in this image) with exactly the calls the reference's and the drop-in's decoders make:
`av.open(path)` -> container; `container.decode(video=0)` -> frames with `to_ndarray(format="rgb24")` and `to_image()`;
`container.close()`.  A "file" is named `fake://<seed>/<n_frames>/<height>x<width>`; its frames are seeded random uint8 images."""
import sys
import types

import numpy as np


class _Frame:
    def __init__(self, arr):
        self._arr = arr

    def to_ndarray(self, format="rgb24"):
        assert format == "rgb24"
        return self._arr

    def to_image(self):
        from PIL import Image
        return Image.fromarray(self._arr)


class _Container:
    def __init__(self, path):
        _, spec = path.split("fake://")
        seed, n, hw = spec.split("/")
        h, w = (int(x) for x in hw.split("x"))
        self._frames = np.random.RandomState(int(seed)).randint(0, 256, (int(n), h, w, 3)).astype(np.uint8)
        self.closed = False

    def decode(self, video=0):
        for f in self._frames:
            yield _Frame(f)

    def close(self):
        self.closed = True


def install():
    m = types.ModuleType("av")
    m.open = lambda path, *a, **k: _Container(str(path))
    sys.modules["av"] = m
    return m


def frames_of(path):
    return _Container(path)._frames
