"""Prints the fields of a bench.py line that a reader checks first."""
import json
import sys

d = json.load(open(sys.argv[1]))
r = d.get("roofline", {})
print("value", d["value"], d["unit"], "| ms/step", d["ms_per_step"])
if r:
    print("self-attn  frac", r["frac"], "us", r["avg_launch_us"], "traffic", r["traffic"])
    print("cross-attn frac", r["cross_attn"]["frac"], "us", r["cross_attn"]["avg_launch_us"], "traffic", r["cross_attn"]["traffic"])
    print("skinny GEMM frac", r["decode_gemm"]["frac"], "us", r["decode_gemm"]["avg_launch_us"], "traffic", r["decode_gemm"]["traffic"])
    print("whole step frac", r["whole_step"]["frac"], "us", r["whole_step"]["us_per_step_incl_encode"])
    print("prefill (cross-attention GEMM) frac", r["prefill"]["frac"], "| forward ms", r["forward"]["ms"], "frac", r["forward"]["frac"])
if "v2_lockstep" in d:
    print("v2_lockstep", d["v2_lockstep"]["tokens_per_s"], "tok/s,", d["v2_lockstep"]["launches_per_step"], "launches/step")
if "cpu_baseline" in d:
    c = d["cpu_baseline"]
    print("cpu_baseline", c["value"], c["unit"], "cores", c["cores"], c.get("thread_calibration_s_per_forward"))
