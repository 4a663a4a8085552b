"""Writes tests/golden/jpeg_fixtures.npz: a few small JPEG files (encoded by Pillow from synthetic pictures) and the RGB arrays
Pillow's own decoder - the reference's decoder, datasets/CustomDataset.py:64 - returns for them.  Data only.

    python tests/golden/make_jpeg_fixtures.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import jpeg_cases as J  # noqa: E402


def main():
    import PIL
    from PIL import features
    picks = dict(J.cases(sizes=((17, 23), (50, 75), (3, 100)), qualities=(75, 100), restarts=(0, 3)))
    picks.update((n, b) for n, b in J.special_cases() if n.startswith(("opt_97", "rows_120")))
    out = {"pillow": np.array(f"{PIL.__version__} / libjpeg {features.version('jpg')} turbo={features.check_feature('libjpeg_turbo')}")}
    for name, data in picks.items():
        out["file_" + name] = np.frombuffer(data, np.uint8)
        out["rgb_" + name] = J.pillow_rgb(data)
    path = os.path.join(HERE, "jpeg_fixtures.npz")
    np.savez_compressed(path, **out)
    print(path, len(picks), "files", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
