# Every profile of a round from ONE box (run through gpurun): bench line, kernel-trace stats of the bench command and of the single-stream
# step, MFMA-busy PMC pass, separate FETCH_SIZE / WRITE_SIZE passes, folded tables + the source hash they belong to.
# usage (on the GPU box): bash tools/prof_all.sh   -> gpurun_out/r4p/*   (copy what is judged into profiles/r04/)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4p; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-also > $O/prof_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ss -- python3 $R/tools/prof_step.py > $O/prof_ss.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-also > $O/pmc_mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_hbm/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-also > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_hbm/write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-also > $O/pmc_w.log 2>&1
cd $R
python3 tools/pmc_mfma.py $O/pmc_mfma $O/final_pmc_mfma.json > $O/pmc_mfma_table.txt
python3 tools/pmc_traffic.py $O/pmc_hbm $O/final_pmc_hbm_traffic.json > /dev/null
python3 -c "import json,sys; sys.path.insert(0,'.'); import bench; json.dump({'csrc_sha16': bench.csrc_sha16(), 'command': 'python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-also'}, open('$O/final_pmc_meta.json','w'))"
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/final_kernel_stats.csv
find $O/prof_ss -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/final_single_stream_kernel_stats.csv
rm -rf $O/prof $O/prof_ss $O/pmc_mfma $O/pmc_hbm
# the bench line last, against THIS pass's counters (bench.py reads profiles/r04/final_pmc_*: refresh the box's copy first)
cp $O/final_pmc_mfma.json $O/final_pmc_hbm_traffic.json $O/final_pmc_meta.json $R/profiles/r04/
python3 $R/bench.py --steps 10 --warmup 3 > $O/bench.log 2>&1
tail -1 $O/bench.log | cut -c1-160
