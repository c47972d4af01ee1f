for rep in 1 2; do
for c in c3 c4; do for w in 4 5; do
  RT_BATCH_WAVES=$w timeout -k 10 200 python bench.py --config $c --steps 4 --warmup 3 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c waves $w', round(d['value']), round(d['ms_per_step'],3))"
done; done; done
for w in 4 5; do
  RT_BATCH_WAVES=$w timeout -k 10 300 python bench.py --config c5 --steps 2 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 waves $w', round(d['value']), round(d['ms_per_step'],3))"
done
for w in 4 5; do echo "== eighth waves $w"; RT_BATCH_WAVES=$w REHEARSAL_WORLDS=1,8 timeout -k 10 300 python tools/partition_rehearsal.py c3 64 10; done
