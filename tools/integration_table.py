"""Prints INTEGRATION.md section 6's table of the reference's README files from a bench.py line (bench_full.json):
python tools/integration_table.py profiles/<tag>_bench_full.json   Not a test."""
import json
import sys

b = json.load(open(sys.argv[1]))
c = b["config_readme_files"]


def f(p):
    return "%.3f / %.3f" % (p[0], p[1])


print("| file | bytes | oracle raw | HIP host raw | oracle framed | HIP host framed | README inMemory raw |")
print("|---|---|---|---|---|---|---|")
for r in c["files"]:
    print("| `%s` | %d | %s | %s | %s | %s | %s |" % (r["file"], r["bytes"], f(r["oracle_raw"]), f(r["hip_host_raw"]),
          f(r["oracle_framed"]), f(r["hip_host_framed"]), f(r["reference_README_inMemory"]["raw"])))
s = c.get("state_38_9MB")
if s:
    print(json.dumps(s))
a = b["config1_alice29"]
print("alice29:", a["hip_host_api_raw"], a["hip_host_api_framed"], "oracle", a["oracle_inMemory_raw"], a["oracle_inMemory_framed"])
print("host_api:", json.dumps(b["host_api"]))
