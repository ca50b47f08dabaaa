#!/usr/bin/env python3
"""The two pixrun_kernel launches of a config-2 sampler step, reported SEPARATELY: rocprofv3's --stats row averages a 7-op launch
(init_conv and the two 8x8 ConvNeXt blocks in front of the 8x8 attention block) with a 70-odd-op launch (everything behind it).

    rocprofv3 --kernel-trace -d DIR -o t --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    python3 tools/pix_launch_split.py DIR

The dispatches of the kernel alternate short / long inside the sampling loop; they are split at the midpoint between the two clusters."""
import csv, glob, os, statistics, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "pixrun_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows.sort()
d = [x[1] / 1e3 for x in rows]
if not d:
    sys.exit("no pixrun_kernel dispatches in the trace")
cut = (min(d) + max(d)) / 2
short = [x for x in d if x < cut]; long_ = [x for x in d if x >= cut]
print(f"pixrun_kernel dispatches: {len(d)}  (average of all {statistics.mean(d):.1f} us)")
for name, v in (("launch 1 (init_conv as im2col + 1x1 conv, the two 8x8 ConvNeXt blocks)", short), ("launch 2 (every level behind the 8x8 attention block, the last conv, the EM update)", long_)):
    if v:
        print(f"  {name}: {len(v)} dispatches, mean {statistics.mean(v):.1f} us, median {statistics.median(v):.1f}, min {min(v):.1f}, max {max(v):.1f}")
print(f"  per SDE step (one of each): {statistics.mean(short) + statistics.mean(long_):.1f} us" if short and long_ else "")
