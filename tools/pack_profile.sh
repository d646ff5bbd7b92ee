#!/bin/bash
# phase profile of the packed schedule (a -DHMX_PACK_PROFILE build of the library): tools/pack_profile.sh "frames plans [ENV=V ...]" ...
cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  set -- $spec
  F=$1; P=$2; shift 2
  echo "== $F pictures, $P plans, $*"
  env HMX_LIB_PATH=$GRAFT_REPO_ROOT/thevc_amd/libhmx_prof.so "$@" timeout -k 10 300 python3 bench.py --frames $F --plans $P --distinct 4 --steps 2 --warmup 1 --no-cpu-baseline --no-ra 2>&1 | grep -E "pack profile|value" | tail -2 | cut -c1-400
done
