import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("value %.3e ms_per_step %.2f init_ms %.1f chol avg_ms %.3f | parity %.2e | stages %d supernodes %d"%(d["value"], d["ms_per_step"], d.get("initialize_only_ms", 0.0), d["kernel_groups"]["cholesky"]["avg_ms"], d["parity"]["max_rel_chi2_diff_vs_cpu"] if d["parity"] else -1, d["structure"]["stages"], d["structure"]["supernodes"]))
ks=sorted(((v["total_ms"],k,v["avg_ms"],v["launches"]) for k,v in d["kernels"].items()),reverse=True)
for t,k,a,n in ks[:12]: print("  %-24s total %7.3f ms  avg %7.1f us  x%d"%(k,t,a*1e3,n))
