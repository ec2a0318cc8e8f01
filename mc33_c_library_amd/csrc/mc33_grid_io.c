/* mc33_grid_io.c -- host-side grid file readers of the reference API (reference
 * include/marching_cubes_33.h:263-311, source/MC33_util_grd.c:171-576).  Plain C, no GPU involved: they fill a
 * _GRD whose rows come from alloc_F; create_MC33 uploads it.  tests/test_grid_io.py reads the same files with
 * the reference's readers and compares every field and sample.
 *
 * Formats
 *   read_grd         DMol ".grd" text: title line, a line that is skipped, "La Lb Lc alpha beta gamma",
 *                    "Na Nb Nc" (intervals), "order xmin xmax ymin ymax zmin zmax" (order 1: x fastest, 3: y
 *                    fastest), then (Na+1)(Nb+1)(Nc+1) values.  d = L/N, r0 = min*d; angles != 90 make the
 *                    grid inclined (_A = unit-edge cell matrix, a along x, b in the xy plane; A_ its inverse).
 *   read_grd_binary  int32 "_GRD" 0x4452475f, int32 title length (<= 159), title, N[3] u32, L[3] f32,
 *                    r0[3] f64, d[3] f64, int32 nonortho, [Ang[3] f32, _A[9] f64, A_[9] f64], rows.
 *   read_scanfiles   "<name><number>", "<name><number+1>", ...: one res x res slice of u16 per file (bytes
 *                    swapped if `order`), first file = top slice.
 *   read_raw_file    bare samples, x fastest: |byte| = 1, 2, 4 integers or 4, 8 floats; byte < 0 = big endian.
 *   read_dat_file    u16 nx, ny, nz then u16 samples, top slice first (TU Wien volume data sets).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/marching_cubes_33.h"

#ifndef GRD_ORTHOGONAL
static void identity3(double (*A)[3]) {
	memset(A, 0, 9 * sizeof(double));
	A[0][0] = A[1][1] = A[2][2] = 1.0;
}
#endif

/* a _GRD with the defaults every reader starts from: unit spacing, origin 0, orthogonal */
static _GRD *blank_grid(void) {
	_GRD *Z = (_GRD *)calloc(1, sizeof(_GRD));
	if (!Z)
		return 0;
	Z->internal_data = 1;
	Z->d[0] = Z->d[1] = Z->d[2] = 1.0;
#ifndef GRD_ORTHOGONAL
	identity3(Z->_A);
	identity3(Z->A_);
#endif
	return Z;
}

static void set_intervals(_GRD *Z, unsigned int nx, unsigned int ny, unsigned int nz) { /* N = L = points - 1 */
	Z->N[0] = nx; Z->N[1] = ny; Z->N[2] = nz;
	Z->L[0] = (float)nx; Z->L[1] = (float)ny; Z->L[2] = (float)nz;
}

static uint16_t swap16(uint16_t v) { return (uint16_t)(v >> 8 | v << 8); }
static uint32_t swap32(uint32_t v) { return v >> 24 | (v >> 8 & 0xFF00u) | (v << 8 & 0xFF0000u) | v << 24; }
static uint64_t swap64(uint64_t v) { return (uint64_t)swap32((uint32_t)v) << 32 | swap32((uint32_t)(v >> 32)); }

/* ------------------------------------------------------------------------------------------------------- */
_GRD *read_grd(const char *filename) { /* UTIL:181-263 */
	FILE *f = fopen(filename, "r");
	if (!f)
		return 0;
	_GRD *Z = blank_grid();
	char line[128];
	float ang[3] = {90.f, 90.f, 90.f};
	int order = 0, lo[3] = {0, 0, 0};
	int ok = Z != 0;
	if (ok) {
		ok = fgets(Z->title, 159, f) && fgets(line, 60, f) && fgets(line, 60, f);
		if (ok) sscanf(line, "%f %f %f %f %f %f", &Z->L[0], &Z->L[1], &Z->L[2], &ang[0], &ang[1], &ang[2]);
		ok = ok && fgets(line, 60, f);
		if (ok) sscanf(line, "%u %u %u", &Z->N[0], &Z->N[1], &Z->N[2]);
		ok = ok && fgets(line, 60, f);
		if (ok) sscanf(line, "%d %d %*d %d %*d %d %*d", &order, &lo[0], &lo[1], &lo[2]);
		ok = ok && Z->N[0] >= 2 && Z->N[1] >= 2 && Z->N[2] >= 2 && (order == 1 || order == 3);
	}
	if (!ok) {
		fclose(f);
		free(Z);
		return 0;
	}
	for (int i = 0; i != 3; i++) {
		Z->d[i] = Z->L[i] / Z->N[i];
		Z->r0[i] = lo[i] * Z->d[i];
	}
	Z->periodic = (lo[0] == 0) | (lo[1] == 0) << 1 | (lo[2] == 0) << 2;
#ifndef GRD_ORTHOGONAL
	for (int i = 0; i != 3; i++)
		Z->Ang[i] = ang[i];
	if (ang[0] != 90 || ang[1] != 90 || ang[2] != 90) { /* cell matrix of a triclinic cell with unit edges (UTIL:218-236) */
		const double rad = 3.14159265358979323846 / 180.0;
		const double ca = cos(ang[0] * rad), cb = cos(ang[1] * rad), gam = ang[2] * rad, sg = sin(gam), cg = cos(gam);
		const double p = ca - cb * cg;                                      /* (b x-y part of c) * sin(gamma)   */
		const double q = sqrt(sg * sg + 2 * ca * cb * cg - ca * ca - cb * cb); /* cell volume / (a b c)          */
		const double isg = 1.0 / sg, iq = 1.0 / q;
		Z->nonortho = 1;
		Z->_A[0][1] = cg;      Z->_A[0][2] = cb;
		Z->_A[1][1] = sg;      Z->_A[1][2] = p * isg;
		Z->_A[2][2] = q * isg;
		Z->A_[1][1] = isg;     Z->A_[0][1] = -cg * isg;
		Z->A_[0][2] = (cg * p - ca * sg * sg) * isg * iq;
		Z->A_[1][2] = -p * isg * iq;
		Z->A_[2][2] = sg * iq;
	}
#endif /* GRD_ORTHOGONAL builds read the samples of an inclined cell as if it were orthogonal, like the reference */
	if (alloc_F(Z)) {
		fclose(f);
		free_memory_grd(Z);
		return 0;
	}
	const unsigned int nx = Z->N[0] + 1, ny = Z->N[1] + 1, nz = Z->N[2] + 1;
	for (unsigned int k = 0; k != nz; k++) { /* values in file order; a short file leaves the rest untouched */
		const unsigned int outer = order == 1 ? ny : nx, inner = order == 1 ? nx : ny;
		for (unsigned int a = 0; a != outer; a++)
			for (unsigned int b = 0; b != inner; b++) {
#ifdef INTEGER_GRD
				double v = 0; /* the reference scans "%f" into the integer sample (undefined); here: value, converted */
				const int got = fscanf(f, "%lf", &v);
#elif GRD_TYPE_SIZE == 8
				double v = 0;
				const int got = fscanf(f, "%lf", &v);
#else
				float v = 0; /* decimal -> float in one rounding, as the reference's "%f" */
				const int got = fscanf(f, "%f", &v);
#endif
				if (got == 1) {
					if (order == 1) Z->F[k][a][b] = (GRD_data_type)v;
					else Z->F[k][b][a] = (GRD_data_type)v;
				}
			}
	}
	fclose(f);
	return Z;
}

/* ------------------------------------------------------------------------------------------------------- */
_GRD *read_grd_binary(const char *filename) { /* UTIL:267-317 */
	FILE *f = fopen(filename, "rb");
	if (!f)
		return 0;
	uint32_t word[2] = {0, 0};
	if (fread(word, 4, 2, f) != 2 || word[0] != 0x4452475fu /* "_GRD" */ || word[1] > 159) {
		fclose(f);
		return 0;
	}
	_GRD *Z = blank_grid();
	if (!Z) {
		fclose(f);
		return 0;
	}
	int32_t inclined = 0;
	size_t got = word[1] ? fread(Z->title, word[1], 1, f) : 1;
	got += fread(Z->N, sizeof Z->N, 1, f);
	got += fread(Z->L, sizeof Z->L, 1, f);
	got += fread(Z->r0, sizeof Z->r0, 1, f);
	got += fread(Z->d, sizeof Z->d, 1, f);
	got += fread(&inclined, sizeof inclined, 1, f);
#ifndef GRD_ORTHOGONAL
	Z->nonortho = inclined;
	if (inclined) {
		got += fread(Z->Ang, sizeof Z->Ang, 1, f);
		got += fread(Z->_A, sizeof Z->_A, 1, f);
		got += fread(Z->A_, sizeof Z->A_, 1, f);
		mult_Abf = _multA_bf; /* the matrices of a file need not be triangular (UTIL:297) */
	}
#else
	if (inclined) { /* skip Ang, _A, A_ (UTIL:304-307) */
		got += fseek(f, 3 * (long)sizeof(float) + 18 * (long)sizeof(double), SEEK_CUR) == 0 ? 3 : 0;
	}
#endif
	Z->periodic = Z->r0[0] == 0 && Z->r0[1] == 0 && Z->r0[2] == 0;
	if (got != (inclined ? 9u : 6u) || alloc_F(Z)) {
		fclose(f);
		if (got != (inclined ? 9u : 6u)) { free(Z); return 0; }
		free_memory_grd(Z);
		return 0;
	}
	for (unsigned int k = 0; k <= Z->N[2]; k++)
		for (unsigned int j = 0; j <= Z->N[1]; j++)
			if (fread(Z->F[k][j], ((size_t)Z->N[0] + 1) * sizeof(GRD_data_type), 1, f) != 1)
				memset(Z->F[k][j], 0, ((size_t)Z->N[0] + 1) * sizeof(GRD_data_type)); /* truncated file */
	fclose(f);
	return Z;
}

/* ------------------------------------------------------------------------------------------------------- */
_GRD *read_scanfiles(const char *filename, unsigned int res, int order) { /* UTIL:326-410 */
	if (!filename || !*filename || res < 2)
		return 0;
	/* split "<stem><digits>" */
	size_t stem = strlen(filename);
	while (stem && filename[stem - 1] >= '0' && filename[stem - 1] <= '9')
		stem--;
	unsigned int number = (unsigned int)atoi(filename + stem);
	char *name = (char *)malloc(stem + 16);
	_GRD *Z = blank_grid();
	if (!name || !Z) {
		free(name);
		free(Z);
		return 0;
	}
	memcpy(name, filename, stem);
	GRD_data_type ***planes = 0;
	unsigned int count = 0, room = 0;
	uint16_t *slice = (uint16_t *)malloc((size_t)res * res * sizeof(uint16_t));
	for (; slice; number++) {
		sprintf(name + stem, "%u", number);
		FILE *f = fopen(name, "rb");
		if (!f)
			break;
		memset(slice, 0, (size_t)res * res * sizeof(uint16_t));
		size_t n = fread(slice, sizeof(uint16_t), (size_t)res * res, f);
		(void)n;
		fclose(f);
		if (count == room) {
			GRD_data_type ***grown = (GRD_data_type ***)realloc(planes, ((size_t)room + 64) * sizeof(void *));
			if (!grown)
				break;
			planes = grown;
			room += 64;
		}
		GRD_data_type **rows = (GRD_data_type **)calloc(res, sizeof(void *));
		int full = rows != 0;
		for (unsigned int j = 0; full && j != res; j++)
			full = (rows[j] = (GRD_data_type *)malloc((size_t)res * sizeof(GRD_data_type))) != 0;
		if (!full) {
			for (unsigned int j = 0; rows && j != res; j++) free(rows[j]);
			free(rows);
			break;
		}
		for (unsigned int j = 0; j != res; j++)
			for (unsigned int i = 0; i != res; i++) {
				const uint16_t v = slice[(size_t)j * res + i];
				rows[j][i] = (GRD_data_type)(order ? swap16(v) : v);
			}
		planes[count++] = rows;
	}
	free(slice);
	free(name);
	if (!count) { /* no file at all: nothing to show (the reference returns a grid with N[2] = -1) */
		free(planes);
		free(Z);
		return 0;
	}
	/* The first file is the top slice: the stack is turned over.  As in the reference (UTIL:398-403) the
	 * number of exchanged pairs is (count-1)/2, so with an even number of files the middle pair stays. */
	for (unsigned int a = 0, last = count - 1; a != last >> 1; a++) {
		GRD_data_type **t = planes[a];
		planes[a] = planes[last - a];
		planes[last - a] = t;
	}
	Z->F = planes;
	set_intervals(Z, res - 1, res - 1, count - 1);
	return Z;
}

/* ------------------------------------------------------------------------------------------------------- */
_GRD *read_raw_file(const char *filename, unsigned int *N, int byte, int isfloat) { /* UTIL:418-520 */
	const int width = abs(byte), big = byte < 0;
	if (isfloat ? (width != 4 && width != 8) : (width != 1 && width != 2 && width != 4))
		return 0;
	if (!N || !N[0] || !N[1] || !N[2])
		return 0;
	FILE *f = fopen(filename, "rb");
	if (!f)
		return 0;
	_GRD *Z = blank_grid();
	if (!Z) {
		fclose(f);
		return 0;
	}
	set_intervals(Z, N[0] - 1, N[1] - 1, N[2] - 1);
	if (alloc_F(Z)) {
		fclose(f);
		free_memory_grd(Z);
		return 0;
	}
	unsigned char *row = (unsigned char *)malloc((size_t)N[0] * width);
	for (unsigned int k = 0; row && k != N[2]; k++)
		for (unsigned int j = 0; j != N[1]; j++) {
			memset(row, 0, (size_t)N[0] * width);
			size_t n = fread(row, width, N[0], f);
			(void)n;
			for (unsigned int i = 0; i != N[0]; i++) {
				const unsigned char *p = row + (size_t)i * width;
				GRD_data_type out;
				if (width == 1)
					out = (GRD_data_type)p[0];
				else if (width == 2) {
					uint16_t v;
					memcpy(&v, p, 2);
					out = (GRD_data_type)(big ? swap16(v) : v);
				} else if (width == 4) {
					uint32_t v;
					memcpy(&v, p, 4);
					if (big) v = swap32(v);
					if (isfloat) {
						float x;
						memcpy(&x, &v, 4);
						out = (GRD_data_type)x;
					} else
						out = (GRD_data_type)v;
				} else {
					uint64_t v;
					double x;
					memcpy(&v, p, 8);
					if (big) v = swap64(v);
					memcpy(&x, &v, 8);
					out = (GRD_data_type)x;
				}
				Z->F[k][j][i] = out;
			}
		}
	free(row);
	fclose(f);
	return Z;
}

/* ------------------------------------------------------------------------------------------------------- */
_GRD *read_dat_file(const char *filename) { /* UTIL:526-576 */
	FILE *f = fopen(filename, "rb");
	if (!f)
		return 0;
	uint16_t dim[3] = {0, 0, 0};
	_GRD *Z = fread(dim, sizeof(uint16_t), 3, f) == 3 && dim[0] && dim[1] && dim[2] ? blank_grid() : 0;
	if (!Z) {
		fclose(f);
		return 0;
	}
	set_intervals(Z, dim[0] - 1u, dim[1] - 1u, dim[2] - 1u);
	if (alloc_F(Z)) {
		fclose(f);
		free_memory_grd(Z);
		return 0;
	}
	uint16_t *row = (uint16_t *)malloc((size_t)dim[0] * sizeof(uint16_t));
	for (unsigned int k = dim[2]; row && k-- != 0;) /* the file starts with the top slice */
		for (unsigned int j = 0; j != dim[1]; j++) {
			memset(row, 0, (size_t)dim[0] * sizeof(uint16_t));
			size_t n = fread(row, sizeof(uint16_t), dim[0], f);
			(void)n;
			for (unsigned int i = 0; i != dim[0]; i++)
				Z->F[k][j][i] = (GRD_data_type)row[i];
		}
	free(row);
	fclose(f);
	return Z;
}
