cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r5j; mkdir -p $O
rm -f gpurun_out/parity_achieved_error.jsonl
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log; tail -5 $O/tests.log
cp gpurun_out/parity_achieved_error.jsonl $O/ 2>/dev/null
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
