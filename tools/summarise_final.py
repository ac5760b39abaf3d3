import json, csv
d = json.loads(open("gpurun_out/bench_final.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["roofline"]["traffic"], d["extras"]["ski"],
      d["extras"]["cached_k"]["mvm_ms"], d["extras"]["block_T11_ms"])
for r in list(csv.DictReader(open("gpurun_out/prof_final/bench_kernel_stats.csv")))[:2]:
    print(r["Name"][:50], r["Calls"], r["AverageNs"])
for l in open("gpurun_out/solve_final.jsonl"):
    d = json.loads(l)
    print(d["config"][:48], "step %.1f ms" % (d["train_step_s"] * 1e3), "iters", d["cg_iters_per_step"], "mean_pred %.3f s" % d["mean_pred_s"],
          "full_pred %.2f s" % d.get("full_pred_s", float("nan")), "rmse %.3f" % d["test_rmse"])
