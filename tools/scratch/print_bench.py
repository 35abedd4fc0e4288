import json, sys
d = json.loads(sys.stdin.read()); r = d["regimes"]
print(sys.argv[1], round(d["value"] / 1e6, 1), "M/s", round(d["ms_per_step"] * 1e3, 1), "us; kernel", round(d["roofline"]["kernel_ms"] * 1e3, 1),
      "full", round(r["fresh"]["full_forward_samples_per_s"] / 1e6, 1), "stress", round(r["stress"]["value"] / 1e6, 1),
      round(r["stress"]["full_forward_samples_per_s"] / 1e6, 1), "frac", round(d["roofline"]["frac"], 3))
