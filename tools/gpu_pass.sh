#!/bin/bash
# One GPU pass under gpurun:  gpurun --timeout T -- 'bash tools/gpu_pass.sh <tag> <step> [<step> ...]'
# Steps (each writes under gpurun_out/<tag>/ and prints a short digest to stdout, which is what gpurun hands back):
#   pytest:<expr>      python -m pytest -m gpu -q <expr words, '+' for spaces>     e.g.  pytest:tests/test_gpu_kernels.py+-k+ragged%or%kv_move
#   bench:<name>:<args>  python bench.py <args, '+' for spaces>  -> <name>.json     e.g.  bench:default:   bench:1p5b:--workload+1p5b
#   trace:<name>:<args>  rocprofv3 --kernel-trace of bench.py <args> + tools/rocprof_summary.py -> <name>_top.txt, _kernel_stats.csv, _gaps.txt
#   pmc:<name>:<counters>:<args>  rocprofv3 --pmc <counters> --kernel-trace of bench.py <args> (its own pass, no other trace domain)
#   py:<script>:<args>   python <script> <args>  (stdout -> <script basename>.log)
#   sh:<name>:<command>  bash -c '<command, + for spaces>'  (stdout -> <name>.out), e.g. a torchrun launch of bench.py
#   smoke              python __graft_entry__.py smoke
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; shift
O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R; export TMPDIR=/tmp
Q="--no-cpu-baseline --no-eager-baseline --skip-extra"
for step in "$@"; do
  kind=${step%%:*}; rest=${step#*:}
  case $kind in
    pytest) IFS='+' read -ra ARGS <<< "$rest"; for i in "${!ARGS[@]}"; do ARGS[$i]=${ARGS[$i]//%/ }; done      # '%' = a space INSIDE one argument (-k+a%or%b)
       n=$(echo "$rest" | tr -c 'A-Za-z0-9_' '_' | cut -c1-60)
       timeout ${VV_STEP_TIMEOUT:-1500} python -m pytest -m gpu -q ${VV_PYTEST_X--x} "${ARGS[@]}" > $O/pytest_$n.log 2>&1; echo "[pytest ${ARGS[*]}] rc=$?"; tail -n ${VV_TAIL:-15} $O/pytest_$n.log;;
    bench) name=${rest%%:*}; a=${rest#*:}; a=${a//+/ }
       timeout ${VV_STEP_TIMEOUT:-900} python bench.py $a > $O/$name.json 2> $O/$name.err; echo "[bench $name] rc=$?"; tail -c 300 $O/$name.err; head -c 1500 $O/$name.json; echo;;
    trace) name=${rest%%:*}; a=${rest#*:}; a=${a//+/ }
       timeout ${VV_STEP_TIMEOUT:-600} rocprofv3 --kernel-trace -d $O/p_$name -o t -- python bench.py $a $Q > $O/${name}_under_rocprof.json 2> $O/${name}_rocprof.err; echo "[trace $name] rc=$?"
       python tools/rocprof_summary.py $O/p_$name/t_results.db $O/$name > $O/${name}_top.txt 2>&1; rm -rf $O/p_$name; head -25 $O/${name}_top.txt;;
    pmc) name=${rest%%:*}; r2=${rest#*:}; ctr=${r2%%:*}; a=${r2#*:}; a=${a//+/ }; ctr=${ctr//+/ }
       timeout ${VV_STEP_TIMEOUT:-600} rocprofv3 --pmc $ctr --kernel-trace -d $O/c_$name -o t -- python bench.py $a $Q > $O/${name}_under_pmc.json 2> $O/${name}_pmc.err; echo "[pmc $name] rc=$?"
       python tools/rocprof_summary.py --pmc $O/c_$name/t_results.db $O/$name > $O/${name}_pmc_top.txt 2>&1; rm -rf $O/c_$name; head -25 $O/${name}_pmc_top.txt;;
    py) sc=${rest%%:*}; a=${rest#*:}; [ "$a" = "$rest" ] && a=""; a=${a//+/ }; n=$(basename $sc .py)
       timeout ${VV_STEP_TIMEOUT:-900} python $sc $a > $O/$n.log 2> $O/$n.err; echo "[py $sc] rc=$?"; tail -n ${VV_TAIL:-30} $O/$n.log; tail -c 600 $O/$n.err;;
    sh) name=${rest%%:*}; a=${rest#*:}; a=${a//+/ }
       timeout ${VV_STEP_TIMEOUT:-900} bash -c "$a" > $O/$name.out 2> $O/$name.err; echo "[sh $name] rc=$?"; tail -c ${VV_TAILC:-1500} $O/$name.out; echo; tail -c 400 $O/$name.err;;
    smoke) timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "[smoke] rc=$?"; tail -3 $O/smoke.log;;
    *) echo "unknown step $step";;
  esac
done
