O=gpurun_out/${TAG:-s3_f384}; mkdir -p $O
for rep in 1 2; do for F in ${FS:-384 448}; do for v in seam ho; do
  if [ $v = ho ]; then export PSDR_SEG_HANDOFF_MIN=1; else unset PSDR_SEG_HANDOFF_MIN; fi
  PSDR_LIB=/root/repo/build/variants/libpsdr_tuning.so python bench.py --workload cfg3 --batch $F --no-extra --no-cpu-baseline --no-post-chain 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['path']['kernels']; print(json.dumps({'v':'$v','F':$F,'value':d['value'],'ms':d['ms_per_step'],'p1_us':k['fft_pass1'].get('device_clock_us_median'),'p2_us':k['fft_pass2'].get('device_clock_us_median')}))" >> $O/bench.jsonl
done; done; done; cat $O/bench.jsonl
