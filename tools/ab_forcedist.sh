for n in 262144 1048576; do for lib in prev tree prev tree; do
  if [ $lib = prev ]; then export NB_ENGINE_LIB=$PWD/tools/ab/prev/nbody3d-webgpu_amd/csrc/libnbody3d_hip.so; else unset NB_ENGINE_LIB; fi
  timeout -k 10 300 python bench.py --force-dist --nbodies $n --steps 4 --warmup 2 --no-cpu-baseline --no-check 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['per_rank']
print('$n $lib', d['config']['kernel_variant'], 'force %.3f reduce %.3f ms_per_step %.3f' % (r['force_kernel_avg_ms'], r['sym_reduce_kernel_avg_ms'], d['ms_per_step']))"
done; done
