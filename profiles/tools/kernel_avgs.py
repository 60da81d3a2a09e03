"""Prints the per-kernel average duration from a rocprofv3 *_kernel_stats.csv (names shortened)."""
import signal
signal.signal(signal.SIGPIPE, signal.SIG_DFL)   # quiet under "| head"
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Name"].split("(")[0][-40:]
    print(f"{name:42s} calls {int(r['Calls']):4d} avg {float(r['AverageNs']) / 1e3:9.1f} us")
