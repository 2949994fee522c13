for v in "400 3" "12 3" "400 100" "150 3" "1000 3"; do set -- $v; echo "VB=$1 DB=$2"; BLS381_VB_EAGER=$1 BLS381_DB_EAGER=$2 timeout -k 10 100 python -u tools/gpu_sweep.py 8:0 | tail -1; done
