#!/bin/bash
# A/B of prebuilt libraries on one box: stage times of the headline view.  usage: run.sh name1 name2 ... (profiles/tools/ab/lib_<name>.so)
names=${@:-old new}
for rep in 1 2; do
for v in $names; do
  cp profiles/tools/ab/lib_$v.so vegs_amd/_lib/libvegsrast.so
  echo "== $v"; python bench.py --stages --no-variants --no-cpu-baseline 2>&1 | grep "stage breakdown\|ms_per_step" | sed 's/.*breakdown//' | cut -c1-330
done
done
