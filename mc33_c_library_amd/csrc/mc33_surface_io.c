/* mc33_surface_io.c -- host-side surface files of the reference API (reference
 * include/marching_cubes_33.h:193-222, source/marching_cubes_33.c:128-327): the binary ".sup" container and
 * the text / Wavefront OBJ / ASCII PLY exports.  Plain C, no GPU involved; the bytes written are identical to
 * the reference's for the same `surface` (tests/test_surface_io.py compares files).
 *
 * .sup layout (little endian, this float build): int32 magic ".sup" = 0x7075732e, float iso, int32 nV,
 * int32 nT, T[nT][3] uint32, V[nV][3] float, N[nV][3] float, color[nV] int32.  Files written by a double
 * build of the reference carry the magic ".sud" = 0x6575732e, a double iso and double V; they are read and
 * narrowed to float (MC:128-136, 178-204).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/marching_cubes_33.h"

#define SUP_MAGIC_FLOAT 0x7075732e  /* ".sup" */
#define SUP_MAGIC_DOUBLE 0x6575732e /* ".sud" */
#if GRD_TYPE_SIZE == 8 /* the double build writes ".sud" and converts ".sup" files when reading (MC:128-131) */
#define SUP_MAGIC_OWN SUP_MAGIC_DOUBLE
typedef float other_real;
#else
#define SUP_MAGIC_OWN SUP_MAGIC_FLOAT
typedef double other_real;
#endif

/* one fwrite of `bytes` bytes, reference style: an item count of 1, so a zero-length block "fails" */
static size_t put_block(FILE *f, const void *p, size_t bytes) { return fwrite(p, bytes, 1, f); }
static size_t get_block(FILE *f, void *p, size_t bytes) { return fread(p, bytes, 1, f); }

int write_bin_s(surface *S, const char *filename) { /* MC:139-157 */
	adjustvectorlenght_s(S);
	FILE *f = fopen(filename, "wb");
	if (!f)
		return -1;
	const int32_t head[1] = {SUP_MAGIC_OWN};
	put_block(f, head, sizeof head);
	put_block(f, &S->iso, sizeof(MC33_real));
	put_block(f, &S->nV, sizeof(int32_t));
	put_block(f, &S->nT, sizeof(int32_t));
	put_block(f, S->T, (size_t)S->nT * 3 * sizeof(int32_t));
	put_block(f, S->V, (size_t)S->nV * 3 * sizeof(MC33_real));
	put_block(f, S->N, (size_t)S->nV * 3 * sizeof(float));
	const size_t last = put_block(f, S->color, (size_t)S->nV * sizeof(int32_t));
	fclose(f);
	return last == 1 ? 0 : -1;
}

surface *read_bin_s(const char *filename) { /* MC:159-219 */
	FILE *f = fopen(filename, "rb");
	if (!f)
		return 0;
	int32_t magic = 0;
	get_block(f, &magic, sizeof magic);
	surface *S = (magic == SUP_MAGIC_FLOAT || magic == SUP_MAGIC_DOUBLE) ? (surface *)calloc(1, sizeof(surface)) : 0;
	if (!S) {
		fclose(f);
		return 0;
	}
	const int wide = magic != SUP_MAGIC_OWN; /* written by the build of the other precision: iso and V are converted */
	if (wide) {
		other_real iso = 0;
		get_block(f, &iso, sizeof iso);
		S->iso = (MC33_real)iso;
	} else
		get_block(f, &S->iso, sizeof(MC33_real));
	get_block(f, &S->nV, sizeof(int32_t));
	get_block(f, &S->nT, sizeof(int32_t));
	S->capv = S->nV;
	S->capt = S->nT;
	S->T = (unsigned int(*)[3])malloc((size_t)S->nT * 3 * sizeof(int32_t));
	S->V = (MC33_real(*)[3])malloc((size_t)S->nV * 3 * sizeof(MC33_real));
	S->N = (float(*)[3])malloc((size_t)S->nV * 3 * sizeof(float));
	S->color = (int *)malloc((size_t)S->nV * sizeof(int32_t));
	int ok = S->T && S->V && S->N && S->color;
	if (ok) {
		get_block(f, S->T, (size_t)S->nT * 3 * sizeof(int32_t));
		if (wide) {
			for (unsigned int j = 0; j < S->nV; j++) {
				other_real v[3];
				get_block(f, v, sizeof v);
				S->V[j][0] = (MC33_real)v[0]; S->V[j][1] = (MC33_real)v[1]; S->V[j][2] = (MC33_real)v[2];
			}
		} else
			get_block(f, S->V, (size_t)S->nV * 3 * sizeof(MC33_real));
		get_block(f, S->N, (size_t)S->nV * 3 * sizeof(float));
		ok = get_block(f, S->color, (size_t)S->nV * sizeof(int32_t)) == 1; /* truncated (or empty) file: no surface */
	}
	fclose(f);
	if (!ok) {
		free_surface_memory(S);
		return 0;
	}
	return S;
}

int write_txt_s(surface *S, const char *filename) { /* MC:221-262 */
	adjustvectorlenght_s(S);
	FILE *f = fopen(filename, "w");
	if (!f)
		return -1;
	fprintf(f, "isovalue: %10.5E\n\nVERTICES:\n%d\n\n", S->iso, S->nV);
	for (unsigned int i = 0; i < S->nV; i++)
		fprintf(f, "%9.6f %9.6f %9.6f\n", S->V[i][0], S->V[i][1], S->V[i][2]);
	fprintf(f, "\n\nTRIANGLES:\n%d\n\n", S->nT);
	for (unsigned int i = 0; i < S->nT; i++)
		fprintf(f, "%8d %8d %8d\n", S->T[i][0], S->T[i][1], S->T[i][2]);
	fputs("\n\nNORMALS:\n", f);
	for (unsigned int i = 0; i < S->nV; i++)
		fprintf(f, "%9.6f %9.6f %9.6f\n", S->N[i][0], S->N[i][1], S->N[i][2]);
	fputs("\n\nCOLORS:\n", f);
	for (unsigned int i = 0; i < S->nV; i++)
		fprintf(f, "%d\n", S->color[i]);
	const int tail = fprintf(f, "\nEND\n");
	fclose(f);
	return tail < 5 ? -1 : 0;
}

int write_obj_s(surface *S, const char *filename) { /* MC:264-296; OBJ indices are 1-based */
	adjustvectorlenght_s(S);
	FILE *f = fopen(filename, "w");
	if (!f)
		return -1;
	fprintf(f, "# isovalue: %10.5E\n# VERTICES %d:\n", S->iso, S->nV);
	for (unsigned int i = 0; i < S->nV; i++)
		fprintf(f, "v %f %f %f\n", S->V[i][0], S->V[i][1], S->V[i][2]);
	fputs("# NORMALS:\n", f);
	for (unsigned int i = 0; i < S->nV; i++)
		fprintf(f, "vn %f %f %f\n", S->N[i][0], S->N[i][1], S->N[i][2]);
	fprintf(f, "# TRIANGLES %d:\n", S->nT);
	for (unsigned int i = 0; i < S->nT; i++) {
		const int a = (int)(S->T[i][0] + 1), b = (int)(S->T[i][1] + 1), c = (int)(S->T[i][2] + 1);
		fprintf(f, "f %d//%d %d//%d %d//%d\n", a, a, b, b, c, c);
	}
	const int tail = fprintf(f, "# END");
	fclose(f);
	return tail < 5 ? -1 : 0;
}

int write_ply_s(surface *S, const char *filename, const char *author, const char *object) { /* MC:298-327 */
	adjustvectorlenght_s(S);
	FILE *f = fopen(filename, "w");
	if (!f)
		return -1;
	fprintf(f, "ply\nformat ascii 1.0\ncomment author: %s\ncomment object: %s\n", author ? author : "", object ? object : "");
	fprintf(f, "element vertex %d\n", S->nV);
	fputs("property float x\nproperty float y\nproperty float z\n"
	      "property float nx\nproperty float ny\nproperty float nz\n"
	      "property uchar red\nproperty uchar green\nproperty uchar blue\n", f);
	fprintf(f, "element face %d\nproperty list uchar int vertex_index\nend_header", S->nT);
	for (unsigned int i = 0; i < S->nV; i++) { /* colour bytes in memory order: 0xAABBGGRR little endian = R, G, B */
		const unsigned char *c = (const unsigned char *)(S->color + i);
		fprintf(f, "\n%f %f %f %f %f %f %d %d %d", S->V[i][0], S->V[i][1], S->V[i][2], S->N[i][0], S->N[i][1], S->N[i][2],
		        c[0], c[1], c[2]);
	}
	for (unsigned int i = 0; i < S->nT; i++)
		fprintf(f, "\n3 %d %d %d", S->T[i][0], S->T[i][1], S->T[i][2]);
	const int tail = fprintf(f, "\n");
	fclose(f);
	return tail ? 0 : -1;
}
