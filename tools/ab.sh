for w in $WL; do for so in $SOS $SOS; do
MZD_AB_SO=$so timeout -k 10 120 python tools/ab.py --workload $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$w $so', d['value'], d['ms_per_step'], d['roofline']['achieved'])"; done; done
