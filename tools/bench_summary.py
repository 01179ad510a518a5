import json, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print("gpus", j["n_gpus"], "build ms", round(j["ms_per_step"], 3), j["build_breakdown_ms"],
      "| knn ms", round(j["knn"]["ms_per_batch"], 3), "topk", round(j["knn"]["topk_kernel_ms"], 3),
      "plan", round(j["knn"]["plan_ms"], 3), "| value", round(j["value"] / 1e6, 2), "Mvec/s",
      "knn", round(j["knn"]["value"] / 1e6, 3), "Mq/s", "| roofline", j["roofline"]["bound"], round(j["roofline"]["frac"], 3), "hbm", round(j["roofline"]["hbm"]["frac"], 3))
