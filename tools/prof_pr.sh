cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in 64 128; do
  for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE"; do
    tag=$(echo $grp | cut -d' ' -f1)
    timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/prof_pr/v${v}_$tag -o p -- python tools/run_algo.py pr kron 2 $v > gpurun_out/prof_pr_v${v}_$tag.log 2>&1
  done
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pr/v${v}_kt -o p -- python tools/run_algo.py pr kron 2 $v > gpurun_out/prof_pr_v${v}_kt.log 2>&1
done
python - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob("gpurun_out/prof_pr/*")):
    f=glob.glob(d+"/p_counter_collection.csv")
    if f:
        agg=collections.defaultdict(lambda:collections.defaultdict(float)); n=collections.defaultdict(int)
        for r in csv.DictReader(open(f[0])):
            k=r["Kernel_Name"][:40]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k]+=1
        for k in agg:
            if "pr_pull" in k or "pr_combine" in k: print(d.split("/")[-1], k, n[k], dict(agg[k]))
    f=glob.glob(d+"/p_kernel_stats.csv")
    if f:
        for r in csv.DictReader(open(f[0])):
            if "pr_" in r["Name"]: print(d.split("/")[-1], r["Name"][:40], r["Calls"], r["AverageNs"])
PY
