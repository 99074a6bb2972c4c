cd $GRAFT_REPO_ROOT
run() { # n L ragged adapters lo
  if [ -n "$5" ]; then export KB_RAGGED_LO=$5; else unset KB_RAGGED_LO; fi
  echo -n "n=$1 L=$2 ragged=$3 adapters=$4 lo=$5 : "; timeout -k 10 120 ./tools/kbench $1 $2 $3 $4 12 0 1 | tail -1 | cut -c38-100
}
for A in 0 1; do
run 2500000 600 0 $A
run 2500000 600 1 $A 300
run 2500000 600 1 $A 1
run 1500000 1000 0 $A
run 1500000 1000 1 $A 1
run 750000 2000 0 $A
run 750000 2000 1 $A 1900
run 300000 5000 1 $A 100
run 30000 50000 1 $A 10000
run 3000 500000 1 $A 100000
run 40000000 36 1 $A 20
run 15000000 100 1 $A 95
run 6000000 250 1 $A 200
run 3000000 500 1 $A 450
done
