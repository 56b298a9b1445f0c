# A/B of the in-tree product library against build_tmp/lib<name>.so variants (tools/ab/build_variant.sh), alternating runs of bench.py's
# arrangement:  bash tools/ab/r06_libab.sh "<name> [<name> ...]" [rounds]   -> gpurun_out/libab/
cd $GRAFT_REPO_ROOT; O=gpurun_out/libab; mkdir -p $O
Q="--gpus 1 --steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity --no-roofline --no-configs"
one() { python bench.py $Q 2>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for r in $(seq 1 ${2:-3}); do
  echo "base   $(one)"
  for n in $1; do echo "$n $(SLIDE_HIP_LIB=$PWD/build_tmp/lib$n.so one)"; done
done | tee $O/result_$(echo $1 | tr ' ' '_').txt
