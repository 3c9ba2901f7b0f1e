"""What an fp32 GEMM emulated on the bf16 matrix pipe would cost in accuracy (CPU, numpy; review item "fp32-accurate GEMM on the
bf16 pipe").  x = h + m (+ l), each part a bf16 (round to nearest even) of what the previous parts left; the product keeps the
terms listed.  One shared-MLP layer of the classifier's last stage: 512 rows x K = 512, operand after BatchNorm + ReLU, weights
N(0, 1/K); errors against the fp64 product.  python tools/probes/bf16_split_accuracy.py -> profiles/r04/bf16_split_accuracy.txt"""
import numpy as np


def bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + ((u >> 16) & 1) + 0x7FFF) >> 16) << 16
    return u.astype(np.uint32).view(np.float32)


def split(x, n):
    parts, r = [], x.copy()
    for _ in range(n):
        h = bf16(r)
        parts.append(h)
        r = (r - h).astype(np.float32)
    return parts


def main():
    rng = np.random.default_rng(0)
    m, k, n = 512, 512, 256
    a = np.maximum(rng.standard_normal((m, k)).astype(np.float32), 0)
    w = (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)
    ref = a.astype(np.float64) @ w.astype(np.float64)
    print(f"rows {m}, K {k}, columns {n}; output rms {np.sqrt((ref ** 2).mean()):.3f}")
    e = np.abs(a @ w - ref)
    print(f"fp32 product                          : max |err| {e.max():.2e}  rms {np.sqrt((e ** 2).mean()):.2e}")
    forms = (("2 parts, hh + hm + mh (3 MFMAs)      ", 2, [(0, 0), (0, 1), (1, 0)]),
             ("2 parts, + mm (4 MFMAs)              ", 2, [(0, 0), (0, 1), (1, 0), (1, 1)]),
             ("3 parts, hh hm mh hl lh mm (6 MFMAs) ", 3, [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]),
             ("3 parts, + ml lm (8 MFMAs)           ", 3, [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1), (1, 2), (2, 1)]))
    for name, parts, terms in forms:
        pa, pw = split(a, parts), split(w, parts)
        acc = np.zeros((m, n), np.float32)
        for i, j in reversed(terms):                     # small terms first
            acc = acc + (pa[i] @ pw[j]).astype(np.float32)
        e = np.abs(acc - ref)
        print(f"{name}: max |err| {e.max():.2e}  rms {np.sqrt((e ** 2).mean()):.2e}")


if __name__ == "__main__":
    main()
