for spp in 4 8 16 32 64; do for lanes in 1 2; do
python bench.py --spp $spp --lanes $lanes --steps 6 --warmup 2 --cpu-seconds 0 --skip-kernels --skip-shared 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spp $spp lanes $lanes  %.1f Msamples/s  %.2f ms/step' % (j['value']/1e6, j['ms_per_step']))"
done; done
