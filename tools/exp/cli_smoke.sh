cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O /tmp/cliout
set -x
python clip_fft.py -t "red square" --size 224-224 --samples 1 --steps 10 --model ViT-B/32 --seed 1 --out_dir /tmp/cliout/a 2>&1 | tail -3
python clip_fft.py -t "red square" --size 1280-720 --samples 200 --steps 40 --seed 1 --no_save --out_dir /tmp/cliout/b 2>&1 | tail -2
python clip_fft.py -t "red square" --size 1280-720 --samples 200 --steps 40 --seed 1 --no_save --precise --out_dir /tmp/cliout/c 2>&1 | tail -2
python clip_fft.py -t "red square" --size 640-360 --samples 50 --steps 20 --seed 1 --no-graph --out_dir /tmp/cliout/d 2>&1 | tail -2
python clip_fft.py -t "red square" --size 640-360 --samples 50 --steps 20 --seed 1 --fast-f16 --dwt --out_dir /tmp/cliout/e 2>&1 | tail -2
python illustrip.py -t "a forest" --size 640-360 --steps 8 --samples 40 --seed 1 --no_save --out_dir /tmp/cliout/f 2>&1 | tail -2
python clip_fft.py -t "x" --size 224-224 --samples 4 --steps 2 --sync 0.5 -i /tmp/none.jpg 2>&1 | tail -2
ls /tmp/cliout/a/* | head -3
