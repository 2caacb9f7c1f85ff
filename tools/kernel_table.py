"""Print the per-kernel table (HIP events, weight gradients on the launch stream) of a bench line:  python tools/kernel_table.py line.json [train|sampler]"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
which = (sys.argv[2] if len(sys.argv) > 2 else "train") + "_step_kernels"
print(d["value"], d["ms_per_step"], d.get("sample_ddpm1000_images_per_sec"))
tot = 0.0
for k in d[which] or []:
    tot += k["ms"]
    print("%-66s %3d %6.3f ms %7.1f us %6.1f TF %6.0f GB/s %-4s %.3f" % (k["kernel"][:66], k["launches"], k["ms"], k["avg_us"], k["tflops"], k["gbs"], k["bound"], k["frac"]))
print("sum %.3f ms" % tot)
