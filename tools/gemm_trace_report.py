"""Per-CU timeline report of a tools/ubench/gemm_sweep `trace` dump ([n_blocks][8] int64 wall-clock stamps at 100 MHz:
0 start, 1 first barrier, 2 main loop end, 5 C tile in LDS, 3 end, 4 HW_ID | XCC_ID << 32)."""
import collections
import sys

import numpy as np


def main(path, show=2):
    a = np.fromfile(path, dtype=np.int64).reshape(-1, 8)
    a = a[a[:, 0] != 0]
    n = len(a)
    t0 = a[:, 0].min()
    st, s1, me, c5, en = [(a[:, i] - t0) / 100.0 for i in (0, 1, 2, 5, 3)]
    hw = a[:, 4] & 0xffffffff
    key = ((a[:, 4] >> 32) & 0xf) * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 10 + ((hw >> 8) & 0xf)
    d = collections.defaultdict(list)
    for i in range(n):
        d[int(key[i])].append(i)
    print(f"{path}: {n} blocks on {len(d)} CUs, span {en.max():.1f} us")
    print(f"  mean prologue {np.mean(s1 - st):.2f}  main {np.mean(me - s1):.2f}  epilogue {np.mean(en - me):.2f} (LDS part {np.mean(c5 - me):.2f})")
    # MFMA-active coverage per CU: union of the main-loop intervals
    cov, gaps = [], []
    for k, idx in d.items():
        iv = sorted((s1[i], me[i]) for i in idx)
        tot, cur_a, cur_b = 0.0, iv[0][0], iv[0][1]
        for x, y in iv[1:]:
            if x > cur_b:
                tot += cur_b - cur_a
                gaps.append(x - cur_b)
                cur_a, cur_b = x, y
            else:
                cur_b = max(cur_b, y)
        tot += cur_b - cur_a
        cov.append(tot)
    print(f"  per-CU time with >= 1 workgroup in its main loop: mean {np.mean(cov):.1f} us of {en.max():.1f}; mid-kernel holes: mean {np.mean(gaps) if gaps else 0:.1f} us x {len(gaps) / len(d):.2f} per CU")
    # overlap: time with 2 WGs in main loop
    both = []
    for k, idx in d.items():
        ev = []
        for i in idx:
            ev += [(s1[i], 1), (me[i], -1)]
        ev.sort()
        c, last, t2 = 0, 0.0, 0.0
        for t, dlt in ev:
            if c >= 2:
                t2 += t - last
            c += dlt
            last = t
        both.append(t2)
    print(f"  per-CU time with 2 workgroups in their main loops: mean {np.mean(both):.1f} us")
    ends = sorted(en)
    print(f"  first block ends at {ends[0]:.1f}, median {ends[n // 2]:.1f}, last {ends[-1]:.1f}; starts of round 2: {sorted(st)[len(d) * 2]:.1f} .. {st.max():.1f}")
    for k in sorted(d)[:show]:
        print("  CU", k)
        for i in sorted(d[k], key=lambda i: st[i]):
            print(f"    blk {i:4d} start {st[i]:7.2f} sync1 {s1[i]:7.2f} mainend {me[i]:7.2f} c5 {c5[i]:7.2f} end {en[i]:7.2f}  main {me[i]-s1[i]:6.2f} epi {en[i]-me[i]:5.2f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2)
