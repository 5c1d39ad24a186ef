B=tools/_bin
for pass in 1 2; do
for sh in "256 128 32640" "256 256 32640" "384 384 32640" "512 256 32640" "512 512 32640"; do
  set -- $sh
  for l in base g16 r12; do
    echo "=== pass $pass $l C $1 CI $2 pixels $3"
    timeout 120 $B/core_bench -r 3 -n 20 -c $1 -i $2 -p $3 $B/$l.so 2>&1 | grep "dcb_nsplit + next" | grep -o "dcb_nsplit + next[^|]*|[^|]*"
  done
done; done
