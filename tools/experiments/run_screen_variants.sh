# usage: run_screen_variants.sh base w3 w4 ...  -- MC throughput (evals/s) of the screening launch per library variant
for v in "$@"; do
  if [ $v = base ]; then unset MI_GNINA_LIB; else export MI_GNINA_LIB=$PWD/gnina_amd/lib/variants/libmi_$v.so; fi
  python tools/screen_demo.py --ligands 1024 --steps 300 --poses 1 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    line=line.strip()
    if line.startswith('{'):
        d=json.loads(line); print('$v', d.get('mc_s'), d.get('mc_evals_per_s'), d.get('ligands_per_s'))
"
done
