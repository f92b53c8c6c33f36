"""Per-step kernel breakdown of the captured full training step from a rocprofv3 kernel trace (tools/profile_c3.sh):
only the last `steps` replays are counted (the first step's MIOpen find phase runs reference convolutions for seconds)."""
import collections
import csv
import glob
import re
import sys

out, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30
f = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
marks = sorted(int(r["Start_Timestamp"]) for r in rows if "k_locate" in r["Kernel_Name"])   # first kernel of the renderer
lo, hi = marks[-steps - 1], marks[-1]
sel = [r for r in rows if lo <= int(r["Start_Timestamp"]) < hi]


def short(n):
    n = re.sub(r"at::native::|\(anonymous namespace\)::|void |dpck::", "", n)
    m = re.search(r"(binary_internal::[A-Za-z]+|[a-z_]+_kernel_cuda|CUDAFunctor_add<[a-z]+>|[A-Za-z_]+Functor<[a-z]+)", n)
    return n.split("<")[0].split("(")[0][:44] + (" " + m.group(1)[:40] if m else "")


agg = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    a = agg[short(r["Kernel_Name"])]
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
busy = sum(v[1] for v in agg.values())
with open(out + "/kernel_stats_per_step.txt", "w") as fh:
    head = ("captured c3 step under rocprofv3 --kernel-trace: %.1f kernels per step, GPU busy %.3f ms of %.3f ms per step "
            "(profiled replays are slower than unprofiled ones; every launch carries ~2 us of tracing)"
            % (len(sel) / steps, busy / steps / 1e6, (hi - lo) / steps / 1e6))
    print(head); fh.write(head + "\n")
    for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        line = "%6.2f%%  %7.1f us/step  %5.1f launches/step  avg %7.1f us  %s" % (100 * v[1] / busy, v[1] / steps / 1e3, v[0] / steps, v[1] / v[0] / 1e3, n)
        print(line); fh.write(line + "\n")
