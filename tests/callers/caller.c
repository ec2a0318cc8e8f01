/* tests/callers/caller.c -- TEST INFRASTRUCTURE: a program written against marching_cubes_33.h (the usage snippet of
 * reference include/marching_cubes_33.h:31-52 plus the helpers the examples call), compiled
 *   (a) against the REFERENCE's header where /root/reference exists, and against this repo's include/ header,
 *   (b) for every GRD_data_type and with / without GRD_ORTHOGONAL,
 * and linked with the product library (or the reference library, for the expected output).
 * It prints the struct layout the header gave it - header drift shows up as a diff between (a)'s two builds - and a
 * digest of the surface it got.  Exit code 3: create_MC33 returned NULL (no GPU). */
#include <stddef.h>
#include <stdio.h>
#include <marching_cubes_33.h>

static double fn(double x, double y, double z) { return 20.0 * (x * x + y * y + z * z); }

static unsigned long long fnv(const void *p, size_t n, unsigned long long h) {
	const unsigned char *b = (const unsigned char *)p;
	for (size_t i = 0; i != n; i++) { h ^= b[i]; h *= 0x100000001b3ull; }
	return h;
}

int main(void) {
	printf("layout GRD %zu F %zu N %zu r0 %zu d %zu L %zu periodic %zu internal_data %zu title %zu\n", sizeof(_GRD), offsetof(_GRD, F),
	       offsetof(_GRD, N), offsetof(_GRD, r0), offsetof(_GRD, d), offsetof(_GRD, L), offsetof(_GRD, periodic), offsetof(_GRD, internal_data),
	       offsetof(_GRD, title));
#ifndef GRD_ORTHOGONAL
	printf("layout GRD Ang %zu nonortho %zu _A %zu A_ %zu\n", offsetof(_GRD, Ang), offsetof(_GRD, nonortho), offsetof(_GRD, _A), offsetof(_GRD, A_));
	printf("layout MC33 _A %zu A_ %zu\n", offsetof(MC33, _A), offsetof(MC33, A_));
#endif
	printf("layout surface %zu T %zu V %zu N %zu color %zu nV %zu nT %zu capt %zu capv %zu iso %zu user %zu\n", sizeof(surface), offsetof(surface, T),
	       offsetof(surface, V), offsetof(surface, N), offsetof(surface, color), offsetof(surface, nV), offsetof(surface, nT), offsetof(surface, capt),
	       offsetof(surface, capv), offsetof(surface, iso), offsetof(surface, user));
	printf("layout MC33 %zu iso %zu memoryfault %zu F %zu O %zu D %zu ca %zu cb %zu nx %zu ny %zu nz %zu store %zu Dx %zu Lz %zu\n", sizeof(MC33),
	       offsetof(MC33, iso), offsetof(MC33, memoryfault), offsetof(MC33, F), offsetof(MC33, O), offsetof(MC33, D), offsetof(MC33, ca), offsetof(MC33, cb),
	       offsetof(MC33, nx), offsetof(MC33, ny), offsetof(MC33, nz), offsetof(MC33, store), offsetof(MC33, Dx), offsetof(MC33, Lz));
	printf("layout types real %zu sample %zu\n", sizeof(MC33_real), sizeof(GRD_data_type));
	_GRD *G = generate_grid_from_fn(-2, -2, -2, 2, 2, 2, 0.05, 0.1, 0.08, fn);
	if (!G) return 2;
	MC33 *M = create_MC33(G);
	if (!M) { puts("create_MC33: NULL"); free_memory_grd(G); return 3; }
	unsigned int cv = 0, ct = 0;
	size_of_isosurface(M, (MC33_real)50.5, &cv, &ct);
	surface *S = calculate_isosurface(M, (MC33_real)50.5);
	if (!S) return 4;
	S->user.p = 0;
	printf("surface nV %u nT %u counted %u %u memoryfault %d iso %g color %08x\n", S->nV, S->nT, cv, ct, M->memoryfault, (double)S->iso,
	       (unsigned)S->color[S->nV - 1]);
	printf("digest T %016llx V %016llx N %016llx\n", fnv(S->T, (size_t)S->nT * 12, 0xcbf29ce484222325ull),
	       fnv(S->V, (size_t)S->nV * 3 * sizeof(MC33_real), 0xcbf29ce484222325ull), fnv(S->N, (size_t)S->nV * 12, 0xcbf29ce484222325ull));
	adjustvectorlenght_s(S);
	printf("after adjust capv %u capt %u first %u\n", S->capv, S->capt, S->T[0][0]);
	free_surface_memory(S); free_MC33(M); free_memory_grd(G);
	return 0;
}
