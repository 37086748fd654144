set -e
python oracle/gen_instances.py rand120 /tmp/r.dat-s; python oracle/gen_instances.py mix4 /tmp/m.dat-s; python oracle/gen_instances.py sdplp40 /tmp/l.dat-s; python oracle/gen_instances.py theta50 /tmp/t.dat-s
for f in r m l t; do for opt in "--dyrankLevel 0" "--dyrankLevel 3" "--highAccMode 1" "--initRho 0.5" "--phase2Tol 1e-7" "--reoptLevel 2 --phase1Tol 1e-2" "--timesLogRank 0.5" "--timesLogRank 0" "--lbfgsListLength 4" "--timeSecLimit 0.01" "--maxADMMIter 5 --phase1Tol 1e-2 --reoptLevel 0"; do
  out=$(timeout -k 5 120 lorads_amd/lib/lorads /tmp/$f.dat-s $opt 2>&1 | grep -E "Primal Objective|End Program|Constraint Violation\(1\)" | tr -s " \t" " " | tr "\n" "|")
  echo "$f [$opt] $out"
done; done
