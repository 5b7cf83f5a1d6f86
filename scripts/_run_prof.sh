set -u
cd $GRAFT_REPO_ROOT
bash scripts/profile_cmd.sh x_cross 'epilogue|kslice' stats,sq,stall_b -- python3 scripts/r6_early_break_probe.py --cases cross > gpurun_out/x_cross.log 2>&1
bash scripts/profile_cmd.sh x_u16000 'epilogue|kslice' stats,sq,stall_b -- python3 scripts/r6_early_break_probe.py --cases u16000 > gpurun_out/x_u16000.log 2>&1
cat gpurun_out/prof_x_cross/summary.md gpurun_out/prof_x_u16000/summary.md
