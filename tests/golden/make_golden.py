"""Regenerates tests/golden/psx_vectors.json from the REFERENCE's own index
generator (utils/src/datagen.cpp compiled unmodified into oracle/_ref by
`make -C oracle ref`).  Runs only in the build container (needs
/root/reference); the JSON it writes is data and travels with the repo.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

CASES = [
    # (num_categories_arg, hot, alpha, shuffle, permute, n_samples, index dtype)
    (999, 8, 1.15, True, True, 6, "int32"),
    (999, 8, 0.0, True, True, 6, "int32"),
    (1023, 8, 1.15, True, True, 4, "int64"),
    (20479, 26, 0.0, True, True, 3, "int32"),
    (20479, 63, 1.05, True, True, 2, "int64"),
    (8, 4, 1.5, False, False, 5, "int32"),
    (50, 5, 1.15, True, False, 5, "int32"),
    (50, 5, 1.15, False, True, 5, "int32"),
    (99999, 16, 2.0, True, True, 3, "int32"),
]


def main():
    O.build(ref=True)
    if O.ref_lib() is None:
        raise SystemExit("oracle/_ref/libref_datagen.so missing (needs /root/reference)")
    out = []
    for n, hot, alpha, shuf, perm, ns, dt in CASES:
        v = O.psx_samples(n, hot, alpha, ns, index=np.dtype(dt).type, shuffle=shuf,
                          permute=perm, use_reference=True)
        out.append(dict(num_categories_arg=n, hot=hot, alpha=alpha, shuffle=shuf,
                        permute=perm, n_samples=ns, index=dt, samples=v.tolist()))
    # one large-shape digest: the C2 generator, first 4096 samples
    big = O.psx_samples(9999999, 64, 1.15, 4096, use_reference=True)
    out.append(dict(num_categories_arg=9999999, hot=64, alpha=1.15, shuffle=True, permute=True,
                    n_samples=4096, index="int32", fnv="%016x" % O.fnv1a64(big),
                    head=big[0][:8].tolist()))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "psx_vectors.json")
    with open(path, "w") as f:
        json.dump(dict(_comment="Outputs of the reference's PowerLawFeatureGenerator "
                                "(utils/src/datagen.cpp), produced by make_golden.py.",
                       cases=out), f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
