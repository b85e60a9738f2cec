for v in 0 1 0 1; do
  echo "DEFERRED_LAST=$v: $(VLP3D_DEFERRED_LAST=$v timeout -k 10 300 python bench.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])")"
done
