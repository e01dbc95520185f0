for M in 0 4; do
for cfg in "32 4096" "16 4096" "64 4096" "8 4096" "4 8192" "8 8192"; do
  set -- $cfg
  for R in 0 512 1024 2048 4096; do python3 tools/ab.py $1 $2 $M $R 0 5 0 | tail -1 | sed "s/^/mode $M pairs $1 size $2 rows $R: /"; done
done; done
