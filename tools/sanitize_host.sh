#!/bin/bash
# Sanitizers on the HOST code, CPU only (GPU AddressSanitizer is not available on this pool): the library's host translation
# units built with clang's AddressSanitizer + UndefinedBehaviorSanitizer and, separately, ThreadSanitizer, and driven through
# the host-only entry point of the Matrix-Market parser on edge-case files (a file ending exactly at a page boundary, short
# files, bad tokens, over-long tokens, no entries, forced thread counts from 1 to 33); and the cgsolver CLI with ASan through
# its no-GPU paths (usage, --cpu, no device, a rank that never answers).  Prints every sanitizer report; exit code 1 if any.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/conjugate-gradient_amd
CL=/opt/rocm/lib/llvm/bin/clang++
W=$(mktemp -d /tmp/cgx_san.XXXX)
cd $W
make -C $P -s build/cgx_kernels.o
python3 - <<'PY'
page = 4096
head = "%%MatrixMarket matrix coordinate real general\n"
n, nz = 9999, 2000
body = "\n".join("%d %d %.17g" % (1 + (7 * k) % n, 1 + (13 * k) % n, 0.5 + k) for k in range(nz))
size = "%d %d %d\n" % (n, n, nz)
pad = (-(len(head) + len(size) + len(body))) % page
pad += page if pad < 2 else 0
open("page.mtx", "w").write(head + "%" + "c" * (pad - 2) + "\n" + size + body)
open("short.mtx", "w").write(head + "4 4 5\n1 1 1\n2 2 2\n3 3\n")
open("bad.mtx", "w").write(head + "4 4 3\n1 1 1\n2 x 2\n3 3 3\n")
open("longtok.mtx", "w").write(head + "4 4 1\n1 1 " + "9" * 300 + "\n")
open("empty.mtx", "w").write(head + "4 4 0\n")
open("tiny.mtx", "w").write(head + "1 1 1\n1 1 2.5")
PY
FILES="$R/tests/golden/lap2D_5pt_n100.mtx page.mtx short.mtx bad.mtx longtok.mtx empty.mtx tiny.mtx nonexistent.mtx"
fail=0
for san in "address,undefined" "thread"; do
  d=obj_${san%%,*}; mkdir -p $d
  for f in cgx_matrix cgx_context cgx_solve cgx_probe cgx_rccl; do
    $CL -x c++ -D__HIP_PLATFORM_AMD__ -fsanitize=$san -fno-omit-frame-pointer -g -O1 -std=c++17 -fPIC -w -I/opt/rocm/include -I$R/include -c $P/csrc/$f.cpp -o $d/$f.o
  done
  $CL -fsanitize=$san -g -O1 -std=c++17 -I$R/include $R/tools/sanitize/parser_harness.cc $d/*.o $P/build/cgx_kernels.o -o harness_${san%%,*} \
      -L/opt/rocm/lib -lamdhip64 -ldl -lpthread -Wl,-rpath,/opt/rocm/lib
  ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 TSAN_OPTIONS=report_signal_unsafe=0 ./harness_${san%%,*} $FILES > out_${san%%,*}.txt 2>&1 || true
  if grep -E "ERROR: AddressSanitizer|runtime error:|WARNING: ThreadSanitizer" out_${san%%,*}.txt; then fail=1; fi
  echo "$san: $(grep -c 'status' out_${san%%,*}.txt) parser calls, $(grep -cE 'ERROR: AddressSanitizer|runtime error:|WARNING: ThreadSanitizer' out_${san%%,*}.txt) sanitizer reports"
done
# the CLI through its no-GPU paths (libcgx.so itself is not instrumented)
g++ -fsanitize=address,undefined -g -O1 -std=c++17 -pthread -I$R/include -o cgsolver_asan $P/host/cg.cc $P/host/cg_main.cc -L$P -lcgx -Wl,-rpath,$P -Wl,-rpath,/opt/rocm/lib
export HIP_VISIBLE_DEVICES=-1 ROCR_VISIBLE_DEVICES=-1 ASAN_OPTIONS=detect_leaks=0
( ./cgsolver_asan; ./cgsolver_asan 64 o.txt --cpu; ./cgsolver_asan 64 o.txt; ./cgsolver_asan 64 o.txt 5 --gpus 2;
  ./cgsolver_asan 64 o.txt 5 --gpus 2 --wireup-timeout 1 --test-hang-stage "device probe:1" ) > out_cli.txt 2>&1 || true
if grep -E "ERROR: AddressSanitizer|runtime error:" out_cli.txt; then fail=1; fi
echo "cgsolver (ASan + UBSan), five no-GPU invocations: $(grep -cE 'ERROR: AddressSanitizer|runtime error:' out_cli.txt) sanitizer reports"
rm -rf $W
exit $fail
