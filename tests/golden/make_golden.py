"""Regenerates the fixtures in this directory.  Needs /root/reference (build container only).

  nttdata0.in/.out, filetest{0..3}.in/.out   verbatim data files of the reference (small)
  nttdata{1,2,3}.npz                         the reference's n = 131072 inputs and expected outputs,
                                             re-encoded as uint32 arrays (the text files are 2 MB each)
  ref_*.npz                                  outputs of the reference's own compiled functions
                                             (oracle/_ref, built from /root/reference by oracle/Makefile)
                                             on seeded inputs, for cases no reference data file covers:
                                             p = 998244353, the CRT path, the CRT primes themselves
"""
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"

from ntt_amd.api import read_in_file, read_out_file  # noqa: E402
from oracle_binding import Oracle, Reference  # noqa: E402


def main():
    shutil.copy(f"{REF}/nttdata/0.in", f"{HERE}/nttdata0.in"); shutil.copy(f"{REF}/nttdata/0.out", f"{HERE}/nttdata0.out")
    for i in range(4):
        shutil.copy(f"{REF}/filetest/{i}.in", f"{HERE}/filetest{i}.in"); shutil.copy(f"{REF}/filetest/{i}.out", f"{HERE}/filetest{i}.out")
    for i in (1, 2, 3):
        n, p, a, b = read_in_file(f"{REF}/nttdata/{i}.in")
        out = read_out_file(f"{REF}/nttdata/{i}.out", n)
        np.savez_compressed(f"{HERE}/nttdata{i}.npz", n=n, p=p, a=a.astype(np.uint32), b=b.astype(np.uint32), out=out.astype(np.uint32))
    ref, orc = Reference(), Oracle()
    assert ref.available, "build oracle/_ref first (make -C oracle)"
    cases = [  # name, n, p, seed, reference function
        ("ref_p998244353_n4096", 4096, 998244353, 0x5EED0001, ref.ntt_multiply),
        ("ref_p998244353_n16384", 16384, 998244353, 0x5EED0002, ref.ntt_multiply),
        ("ref_p1004535809_n2048", 2048, 1004535809, 0x5EED0003, ref.ntt_multiply),
        ("ref_p1224736769_n8192", 8192, 1224736769, 0x5EED0004, ref.ntt_multiply),
        ("ref_p104857601_n64", 64, 104857601, 0x5EED0005, ref.ntt_multiply),
        ("ref_crt_n1024", 1024, 1337006139375617, 0xB16B00B5, ref.crt_ntt_multiply),
        ("ref_crt_n8192", 8192, 1337006139375617, 0xB16B00B6, ref.crt_ntt_multiply),
        ("ref_crt_n4_max", 4, 1337006139375617, None, ref.crt_ntt_multiply),
    ]
    for name, n, p, seed, fn in cases:
        if seed is None:
            a = np.full(n, p - 1, dtype=np.uint64); b = a.copy()
        else:
            a = orc.fill_random(n, p, seed); b = orc.fill_random(n, p, seed ^ 0xFFFF)
        out = fn(a, b, p)
        np.savez_compressed(f"{HERE}/{name}.npz", n=n, p=p, a=a, b=b, out=out)
        print(name, "ok")


if __name__ == "__main__":
    main()
