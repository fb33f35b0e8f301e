#!/bin/bash
# rocprofv3 evidence for one command:  tools/profile.sh <tag> <cmd...>
#   pass 0: --kernel-trace --stats            (per-kernel time)
#   pass 1..: ONE --pmc group per run, with --kernel-trace only (never with other trace
#             domains), every run under its own `timeout` (a TA_* group once hung rocprofv3)
# Outputs under gpurun_out/prof_<tag>/ ; tools/summarize_prof.py writes summary.md
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o p -- "$@" > "$OUT/kt.log" 2>&1
i=0
if [ -n "${PROF_GROUPS_FILTER:-}" ]; then GROUPS_FILTER="$PROF_GROUPS_FILTER"; elif [ "${PROF_SHORT:-0}" = 1 ]; then GROUPS_FILTER='^FETCH_SIZE|^WRITE_SIZE|^SQ_WAVES|^SQ_INSTS_LDS|^TCC_HIT'; else GROUPS_FILTER='.'; fi
while read -r group; do
  echo "$group" | grep -Eq "$GROUPS_FILTER" || continue
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $group --output-format csv -d "$OUT/pmc$i" -o p -- "$@" > "$OUT/pmc$i.log" 2>&1
done <<'GROUPS'
FETCH_SIZE GRBM_GUI_ACTIVE
WRITE_SIZE
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum
GROUPS
grep -h "^{" "$OUT/kt.log" | tail -1 | cut -c1-400
python tools/summarize_prof.py "$OUT" > "$OUT/summary.md" 2>&1
head -40 "$OUT/summary.md"
