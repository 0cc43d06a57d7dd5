"""One-screen summary of a bench.py JSON line (headline mode, the separately reported mode, e2e, CPU rows)."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
def show(tag, m):
    print("%s: chunks/s %s  ms/step %s  roofline %s frac %s (isolated %s)  step %s" % (
        tag, m["value"], m["ms_per_step"], m["roofline"]["kernel"][:40], m["roofline"]["frac"], m["roofline"].get("isolated_frac"), m["step_flops"]))
    for k in m["kernels"]:
        print("  %-40s avg %.4f ms  %.1f TF  share %.3f  isolated %.1f TF" % (k["kernel"][:40], k["avg_ms"], k["tflops"], k["share_of_step"], k.get("isolated_tflops") or 0))
show(d["config"]["precision"], d)
for other in ("f16x3", "f32"):
    if other in d:
        show(other, d[other])
if "e2e" in d:
    print("e2e:", d["e2e"]["value"], "chunks/s", d["e2e"]["ms_per_step"], "ms/step")
if "cpu_baseline" in d:
    print("cpu:", d["cpu_baseline"]["rows"])
if "comm" in d:
    print("comm:", json.dumps(d["comm"])[:600])
