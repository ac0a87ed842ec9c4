python tools/feed_driver.py 1000 feedacq
node tests/js/node_default_workload.js 2000 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('node plain', d['ms_per_frame'])"
node tests/js/node_default_workload.js 2000 feed | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('node feed', d['ms_per_frame'], d['frames_landed_in_loop'])"
node tests/js/node_default_workload.js 2000 feednoacq | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('node feed no acquire', d['ms_per_frame'], d['frames_landed_in_loop'])"
python -m pytest tests/test_c_client.py -m gpu -q 2>&1 | tail -3
