for st in 3 4; do
timeout 900 python bench.py --mode sample --no-cpu --no-roofline --no-secondary --sample-images 1024 --sample-streams $st 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams', d.get('sample_streams'), d.get('sample_ddpm1000_images_per_sec'), d.get('sample_seconds'))"
done
