set -u
H=nbody-demo-2023_amd/host
O=gpurun_out/r04_cli
mkdir -p $O
run() { name=$1; shift; echo "\$ $*" > $O/$name.txt; ( "$@" ) >> $O/$name.txt 2>&1; echo "exit status $?" >> $O/$name.txt; }
run nbody_2000_500 $H/nbody.x 2000 500
run nbody_16384_500 $H/nbody.x 16384 500
run nbody_262144_200 $H/nbody.x 262144 200
run nbody_fp64_262144_100 $H/nbody_fp64.x 262144 100
run nbody_1048576_100_world1 env NBODY_WORLD=1 NBODY_RANK=0 $H/nbody.x 1048576 100
run nbody_1048576_100_8ranks env NBODY_GPUS=8 $H/nbody.x 1048576 100
run nbody_262144_200_8ranks env NBODY_GPUS=8 $H/nbody.x 262144 200
run nbody_v5_16384_100_cpu_gpu_0.25_2gpus env NBODY_GPUS=2 $H/nbody_v5.x 16384 100 cpu+gpu 0.25
run nbody_v5_262144_300_cpu_gpu_tuning_4gpus env NBODY_GPUS=4 $H/nbody_v5.x 262144 300 cpu+gpu -1
run nbody_v5_65536_400_cpu_gpu_tuning_3gpus env NBODY_GPUS=3 $H/nbody_v5.x 65536 400 cpu+gpu -1
run nbody_v5_16384_100_cpu_gpu_one_gpu $H/nbody_v5.x 16384 100 cpu+gpu 0.3
tail -n 12 $O/nbody_v5_262144_300_cpu_gpu_tuning_4gpus.txt $O/nbody_v5_65536_400_cpu_gpu_tuning_3gpus.txt $O/nbody_262144_200_8ranks.txt
