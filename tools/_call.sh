for i in 1 2 3; do
echo "== r5"; python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --dense 0 --sustained 0 2>/dev/null | python tools/benchline.py
echo "== r4"; (cd _r4 && python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --dense 0 --sustained 0 2>/dev/null | python tools/benchline.py)
done
echo "== r5 driver"; python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --dense 0 --sustained 0 2>/dev/null | python tools/benchline.py
echo "== r4 driver"; (cd _r4 && python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --dense 0 --sustained 0 2>/dev/null | python tools/benchline.py)
