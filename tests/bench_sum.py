import json,sys
d=json.load(open(sys.argv[1]))
print("value %.3e ms_per_step %.2f init_ms %.1f chol avg_ms %.3f | parity %.2e | stages %d supernodes %d"%(d["value"], d["ms_per_step"], d["init_ms"], d["kernel_groups"]["cholesky"]["avg_ms"], d["parity"]["max_rel_chi2_diff_vs_cpu"] if d["parity"] else -1, d["structure"]["stages"], d["structure"]["supernodes"]))
