for v in 8 2 4 6 12 100; do echo "rows_at=$v"; IMPGPU_MIX_ROWS_AT=$v python bench.py --mixed 1024 --steps 20 --warmup 10 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['one_call'])"; done
echo nosort; IMPGPU_MIX_NOSORT=1 python bench.py --mixed 1024 --steps 20 --warmup 10 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['one_call'])"
echo 4096; python bench.py --mixed 4096 --steps 10 --warmup 10 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['one_call'], d['launch_per_frame'])"
