set -e
O=gpurun_out/r03f; mkdir -p $O
R=$PWD
timeout -k 10 300 python tools/analysis/fit_phases.py > $O/fit_phases.log 2>&1 || { tail -30 $O/fit_phases.log; exit 1; }
grep -v amdgpu $O/fit_phases.log | tail -40
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d $R/$O/fit_trace -- python3 $R/tools/analysis/fit_phases.py > $R/$O/fit_phases_traced.log 2>&1 || { tail -30 $R/$O/fit_phases_traced.log; exit 1; }
cd $R
python tools/analysis/fit_trace_summary.py $O/fit_trace > $O/fit_trace_summary.log 2>&1 || true
cat $O/fit_trace_summary.log
grep -v amdgpu $O/fit_phases_traced.log | tail -14
ls $O/fit_trace/*/ | head; du -sh $O/fit_trace
rm -rf $O/fit_trace
