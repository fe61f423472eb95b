ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for b in 880 1760 2640 4096; do python bench.py --ragged --no-cpu --steps 5 --batch $b --tlo 2400 --thi 3000 | grep "^{" | cut -c1-150; done
for b in 1200 2400 4096; do python bench.py --ragged --no-cpu --steps 5 --batch $b --tlo 1600 --thi 2400 | grep "^{" | cut -c1-150; done
for b in 2000 4096; do python bench.py --ragged --no-cpu --steps 5 --batch $b --tlo 200 --thi 1600 | grep "^{" | cut -c1-150; done
