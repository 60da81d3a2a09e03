"""GPU busy fraction from a rocprofv3 kernel trace CSV: union of the kernel intervals / (last end - first start), the mean
number of kernels resident at once, and the launch count per second."""
import csv, sys
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # drop the first N launches (warm-up / allocation phase)
iv = iv[skip:]
t0, t1 = iv[0][0], max(e for _, e in iv)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"{len(iv)} launches over {(t1 - t0) / 1e6:.1f} ms: busy {busy / (t1 - t0):.3f}, mean concurrency {sum(e - s for s, e in iv) / (t1 - t0):.2f}, "
      f"{len(iv) / ((t1 - t0) / 1e9):.0f} launches/s")
