# On the GPU box: pipelined step time against the number of batches in flight and the chaining event, two passes
# (MI355_BLS_CHAIN_EV exists only in a build made with BLS_EXTRA_FLAGS=-DBLS_EXPERIMENTS: LIB=<name> selects nim-blscurve_amd/variants/<name>.so)
cd $GRAFT_REPO_ROOT
[ -n "$LIB" ] && export MI355_BLS_LIB=$GRAFT_REPO_ROOT/nim-blscurve_amd/variants/$LIB.so
for pass in 1 2; do
for cfg in "3 -" "2 -" "4 -" "3 hm" "3 clear" "3 sig" "3 lines"; do
  set -- $cfg
  echo -n "inflight=$1 chain_ev=$2  "
  if [ "$2" = "-" ]; then python3 bench.py --steps 30 --warmup 4 --no-cpu --no-aux --no-one-caller --inflight $1 2>/dev/null | python3 -c "import sys,json; print(round(json.loads(sys.stdin.readline())['ms_per_step'],3))"
  else MI355_BLS_CHAIN_EV=$2 python3 bench.py --steps 30 --warmup 4 --no-cpu --no-aux --no-one-caller --inflight $1 2>/dev/null | python3 -c "import sys,json; print(round(json.loads(sys.stdin.readline())['ms_per_step'],3))"; fi
done; done
