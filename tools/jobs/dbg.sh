mkdir -p gpurun_out/dbg
/opt/rocm/bin/rocgdb -batch -ex run -ex "p \$_siginfo._sifields._sigfault.si_addr" -ex "info registers rdi rsi rdx rcx" -ex "x/3i \$pc" -ex "info proc mappings" --args python bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/dbg/gdb.out 2>&1
