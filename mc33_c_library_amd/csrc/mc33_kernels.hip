// mc33_kernels.hip -- HIP kernels (gfx950 / CDNA4, wave64) and the device-level C ABI (include/mc33_hip.h)
// of the MI355X Marching Cubes 33 extractor.  One shared object per grid sample type, like the
// reference's one-type-per-compile model (reference include/marching_cubes_33.h:57-88):
//     default            -> float samples           (libMC33_f32.so)
//     -DMC33_GRD_F64     -> double samples AND double arithmetic / vertices (libMC33_f64.so)
//     -DMC33_GRD_U8 / _U16 / _U32 -> unsigned char / short / int samples (libMC33_u8 / _u16 / _u32.so)
//
// Passes of one extraction, in launch order ("MC:" = reference source/marching_cubes_33.c):
//   k_sweep<S,NI,ZM> - streams the volume once (MC:1832-1868) and does nothing else that costs bandwidth: sign bit per
//                   sample by wave ballot, the bit rows of a 64-row x 256-sample tile slice parked one row per LANE, so
//                   that "is any cell of this slice cut" is a handful of 64-bit logic ops.  For each slice that is cut it
//                   leaves the bit planes (each plane once; 256 bytes in compact form), a header (cut cells, halo bits)
//                   and a partial sum for k_slots.  The halo column comes from the neighbouring wave through an LDS
//                   mailbox.  S: narrow samples are loaded S to a dword; NI: isovalues classified per pass
//                   (mc33hip_sweep_many), each with its own "lane" of output buffers; ZM: how a sample is classified.
//   k_boundary    - the slice between two z-tiles of k_sweep, from the edge planes both left behind.
//   k_slots       - exclusive sums of (cut cells, batches of 64 of them) over the slice slots in storage order.
//   k_cells       - the waves take cut slices off a list k_slots made, 64 cut cells per step: sign index from the bit planes; fast cells (interior,
//                   no test needed, no sample == iso) finished from a 256-entry table, cells whose sign index needs the
//                   face / interior tests tested here (TESTED records), the rest queued for k_slow_plan; one 8-byte work
//                   record per cut cell, contiguous in the reference's visiting order, (#new vertices, #triangles) per
//                   row segment, and one descriptor per batch of 64 records for the vertex pass.
//   k_slow_plan   - the generic plan (MC:683-779, 788-1224: cells on the grid's 0-faces, corners equal to the isovalue).
//   k_slow_count  - triangles of cells with a corner equal to the isovalue, by vertex identity (MC:1235); enqueued when the
//                   context's last extraction had such cells.
//   k_seg_fix     - offsets of the row segments a slow cell changed (and the identity counts k_slow_count was not enqueued for).
//   k_scan_*      - exclusive prefix sums over the row segments in the reference's sweep order: this IS the
//                   reference's vertex / triangle numbering.
//   k_emit_vertices<MODE> - one WAVE per batch of 64 records of one slice slot: the sample rows the batch needs staged in
//                   LDS by cooperative 16-byte loads, one lane per vertex (MC:810-1230, 485-585).
//   k_emit_fast_triangles - one thread per record: vertex ids of the nine shared edges through the owners' records,
//                   triangles (MC:1235-1250); leaves the counters of the extraction in the host's pinned copy.  First of the
//                   emit passes while the record set fits the last-level cache, behind the vertex pass otherwise.
//   k_emit_slow_slots / k_emit_slow - the generic emit of the slow records: 16 lanes per record (a lane per pattern slot) in
//                   sequence behind the fast passes when they are few; a thread per record, on large grids on a second
//                   stream beside the fast passes, when they are many.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <algorithm>
#include <cmath>
#include <limits>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/mc33_hip.h"
#ifdef MC33_GRD_F64
#define MC33_REAL_DOUBLE  // MC33_real is double in the double build (reference marching_cubes_33.h:80-82)
#endif
#if !defined(MC33_GRD_U8) && !defined(MC33_GRD_U16) && !defined(MC33_GRD_U32)
#define MC33_NAN_SAMPLES 1  // float / double grids may hold NaN samples
#else
#define MC33_INT_SAMPLES 1  // unsigned integer samples: no NaN, no signed zero (k_sweep's ZM)
#endif
#include "mc33_cell.h"
#include "mc33_lut_data.h"
#include "mc33_rules_data.h"

using namespace mc33;

#if defined(MC33_GRD_U16)
typedef uint16_t sample_t;
#define MC33_SAMPLE_BYTES 2
#elif defined(MC33_GRD_U8)
typedef uint8_t sample_t;
#define MC33_SAMPLE_BYTES 1
#elif defined(MC33_GRD_U32)
typedef uint32_t sample_t;
#define MC33_SAMPLE_BYTES 4
#elif defined(MC33_GRD_F64)
typedef double sample_t;
#define MC33_SAMPLE_BYTES 8
#else
typedef float sample_t;
#define MC33_SAMPLE_BYTES 4
#endif

// The translation unit in parts (round 5: it had grown to 4 500 lines), included in this order:
#include "mc33_records.hip.h"
#include "mc33_sweep.hip.h"
#include "mc33_tail.hip.h"
#include "mc33_emit.hip.h"
#include "mc33_context.hip.h"
#include "mc33_extract.hip.h"
