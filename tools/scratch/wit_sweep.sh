set -e
run() { echo "== $*"; env "$@" timeout -k 10 200 python tools/gpu_witness_dev.py 2>&1 | grep "proofs/s"; }
run N=16384
run N=24576 BLS381_WIT_CACHE=69 BLS381_WIT_CH=12288
run N=20480 BLS381_WIT_CACHE=83 BLS381_WIT_CH=10240
run N=28672 BLS381_WIT_CACHE=59 BLS381_WIT_CH=14336
run N=16384
