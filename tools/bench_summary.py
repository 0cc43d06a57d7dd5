import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("chunks/s", d["value"], "ms/step", d["ms_per_step"], d["step_flops"])
for k in d["kernels"]:
    print("  %-34s avg %.4f ms  %.1f TF  share %.3f  isolated %.1f TF" % (k["kernel"][:34], k["avg_ms"], k["tflops"], k["share_of_step"], k.get("isolated_tflops") or 0))
