import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d["kernel_classes"]
for n in ("linear_fwd[24576x128x256]", "linear_fwd[24576x128x265]", "linear_dgrad[24576x128x256]", "linear_fwd[24576x256x512]", "linear_dgrad[24576x256x512]", "linear_fwd[24576x512x512]", "linear_dgrad[24576x512x512]", "linear_fwd[24576x693x512]"):
    v = k[n]; print("  %-34s %6.1f us" % (n, v["ms"] / v["launches"] * 1e3))
print("  step", round(d["ms_per_step"], 2))
