# usage: envsweep.sh "VAR=a VAR=b ..."  -> ms/step of the headline bench for each setting
for kv in "$@"; do
  r=$(env $kv python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*')
  echo "$kv -> $r"
done
