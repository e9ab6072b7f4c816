#!/bin/bash
# Same-box A/B of config 3 across the rounds' libraries: each round's own tree (sources + its library, built from the round's
# last commit into .abtrees/<round>, git-ignored) runs ITS tools/ab.py; two interleaved rounds.   bash tools/r5_rounds_ab.sh <tag>
tag=${1:-r5_rounds}
{
for r in 1 2; do
  for t in r3 r4; do (cd .abtrees/$t && python tools/ab.py 2>&1 | grep "us/step" | head -4 | sed "s/^ *default/      round ${t#r} library/"); done
  python tools/ab.py 2>&1 | grep "us/step" | head -4 | sed "s/^ *default/      round 5 library/"
done
} > gpurun_out/${tag}_ab.txt
cat gpurun_out/${tag}_ab.txt
