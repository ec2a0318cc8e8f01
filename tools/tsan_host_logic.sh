#!/bin/bash
# Developer tool (CPU): the host layer of the C API (csrc/mc33_capi.c as it ships) on the fake device layer of the CPU suite
# (tests/host_emu/fake_hip.cpp + emu.cpp), built with -fsanitize=thread and driven by a small C program with z-slabs on four
# "devices" - a thread per device, the colour helper thread, the host block cache.  Prints the surfaces' sizes; any data race
# is reported by ThreadSanitizer on stderr.      tools/tsan_host_logic.sh [MC33_HIP_DEVICES list, default 0,1,2,3,0,1]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
cd "$T"
gcc -O1 -g -fsanitize=thread -ffp-contract=off -std=c11 -fPIC -c "$R/mc33_c_library_amd/csrc/mc33_capi.c" -o capi.o
g++ -O1 -g -fsanitize=thread -ffp-contract=off -std=c++17 -fPIC -shared "$R/tests/host_emu/emu.cpp" "$R/tests/host_emu/fake_hip.cpp" capi.o -o libhl.so -lpthread
cat > drv.c <<'C'
#include <stdio.h>
#include <math.h>
#include "marching_cubes_33.h"
static double fn(double x, double y, double z) { return sin(37 * x) * cos(41 * y) + sin(43 * z) * 0.7 + cos(29 * x * y); }
int main(void) {
	_GRD *G = generate_grid_from_fn(0, 0, 0, 4, 4, 4, 4.0 / 71, 4.0 / 71, 4.0 / 71, fn);
	MC33 *M = create_MC33(G);
	if (!M) { puts("create_MC33 failed"); return 1; }
	for (int k = 0; k < 3; k++) {
		surface *S = calculate_isosurface(M, 0.05f * k);
		printf("iso %g: nV %u nT %u\n", 0.05 * k, S->nV, S->nT);
		free_surface_memory(S);
	}
	unsigned nV, nT;
	size_of_isosurface(M, 0.0f, &nV, &nT);
	printf("size_of_isosurface: nV %u nT %u\n", nV, nT);
	free_MC33(M); free_memory_grd(G);
	return 0;
}
C
gcc -O1 -g -fsanitize=thread -I"$R/include" drv.c -o drv -L. -lhl -lm -Wl,-rpath,.
MC33_HIP_DEVICES=${1:-0,1,2,3,0,1} ./drv
rm -rf "$T"
