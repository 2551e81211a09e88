# per-phase clock of k_jpeg_sync<1> (diagnostic build): tools/jpeg_phase_clock.sh   [on the GPU box; the .so is built in the container]
for n in 34 240; do CAMA_HIP_LIB=$PWD/_ab/libcama_trace.so python tools/jpeg_phase_clock.py $n; done
