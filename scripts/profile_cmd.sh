#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel stats + PMC passes of an ARBITRARY python command, summarised
# for the kernels whose names match a regular expression.  Every pass is its own run of the command
# (--kernel-trace --stats first; then one --pmc pass per counter set, each with --kernel-trace only).
# Usage: scripts/profile_cmd.sh <tag> '<kernel regex>' <passes> -- python3 <script> [args...]
#   passes: comma list of  stats,fetch,write,sq,tcc,stall_a,stall_b   (or "all", or "traffic" = stats,fetch,write)
# Outputs under gpurun_out/prof_<tag>/ ; the summary (summary.md) is what gets copied to profiles/.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
if [ $# -lt 4 ]; then
  echo "usage: $0 <tag> '<kernel regex>' <passes> -- python3 <script> [args...]" >&2
  exit 2
fi
TAG=$1; KERNELS=$2; PASSES=$3; shift 3
[ "${1:-}" = "--" ] && shift
if [ $# -lt 1 ]; then
  echo "$0: no command after --" >&2
  exit 2
fi
# The profiler's preloaded library initialises the GPU before the program starts (with --pmc it does), and on this
# pool a process that has initialised the GPU must not exec another program: what follows `--` has to be the
# interpreter BINARY itself (an ELF file), never env / bash -c / taskset / numactl or a `#!/usr/bin/env` script.
# (the checks look at what the name resolves to; the command runs the name as given -- a symlink is not an exec hop, and a
# virtualenv's interpreter finds its pyvenv.cfg only next to the unresolved path)
ASGIVEN=$(command -v -- "$1" || true)
ASGIVEN=${ASGIVEN:-$1}
PROG=$(readlink -f -- "$ASGIVEN")
case "$(basename -- "$PROG")" in
  env|bash|sh|dash|taskset|numactl|nice|timeout|stdbuf) echo "$0: '$1' would exec the real program after the GPU is initialised: name the interpreter itself" >&2; exit 2;;
esac
if [ ! -x "$PROG" ] || [ "$(head -c 4 -- "$PROG" | od -An -c | tr -d ' ')" != "177ELF" ]; then
  echo "$0: '$1' ($PROG) is not an ELF executable (a script's #! line is an exec hop too)" >&2
  exit 2
fi
shift
set -- "$ASGIVEN" "$@"
[ "$PASSES" = "all" ] && PASSES=stats,fetch,write,sq,tcc,stall_a,stall_b
[ "$PASSES" = "traffic" ] && PASSES=stats,fetch,write
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
declare -A PMC=(
  [fetch]="FETCH_SIZE"
  [write]="WRITE_SIZE"
  [sq]="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
  [tcc]="TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
  [stall_a]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
  [stall_b]="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM"
)
echo "$*" > "$OUT/command.txt"
for P in ${PASSES//,/ }; do
  if [ "$P" = "stats" ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- "$@" > "$OUT/run_stats.log" 2>&1
  else
    rocprofv3 --pmc ${PMC[$P]} --kernel-trace --output-format csv -d "$OUT/$P" -- "$@" > "$OUT/run_$P.log" 2>&1
  fi
  echo "pass $P rc=$?"
done
KERNELS="$KERNELS" python3 scripts/summarize_profile.py "$OUT" > "$OUT/summary.md"
# keep the summary, the kernel stats table and the logs; the raw traces / counter dumps run to hundreds of MB
# (gpurun copies at most 64 MiB back)
cp "$(ls "$OUT"/stats/*/*kernel_stats.csv 2>/dev/null | tail -1)" "$OUT/kernel_stats.csv" 2>/dev/null
for P in ${PASSES//,/ }; do rm -rf "$OUT/$P"; done
cat "$OUT/summary.md"
