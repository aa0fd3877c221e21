import json, sys
d = json.load(open(sys.argv[1]))
print("insert M/s %.1f  ms/step %.2f  query M/s %.1f" % (d["value"] / 1e6, d["ms_per_step"], d["query_value"] / 1e6))
print("stats", d["stats"])
for k, v in (d.get("kernel_classes") or {}).items():
    print(" %-14s launches/step %6.0f  ms/step %8.3f  avg_us %9.1f  alg_GBps %s" % (k, v["launches_per_step"], v["seconds_per_step"] * 1e3, v["avg_launch_us"], v.get("alg_GBps")))
print("roofline", d.get("roofline"))
print("ceiling", d.get("random_access_ceiling"))
print("cpu", d.get("cpu_baseline"))
