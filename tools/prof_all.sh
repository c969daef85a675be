set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2p; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py --steps 10 --warmup 3 > $O/bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/prof_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ss -- python3 $R/tools/prof_step.py > $O/prof_ss.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_hbm/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_hbm/write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_w.log 2>&1
tail -1 $O/bench.log | cut -c1-160
