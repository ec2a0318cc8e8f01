/* tools/uaf_trap.c -- developer aid: LD_PRELOAD this to catch WRITES TO FREED HEAP MEMORY by any thread or
 * library of the process (uninstrumented ones included, which AddressSanitizer cannot see).
 *
 * free(p) does not release p: the block is filled with 0xA5 and parked in a FIFO quarantine together with the
 * backtrace of the free() call.  When a block leaves the quarantine (and for all parked blocks at exit) the fill
 * is verified; a changed byte means somebody wrote through a dangling pointer, and the report names the block,
 * the offset, the bytes found and who freed it.
 *
 *   gcc -O2 -g -shared -fPIC tools/uaf_trap.c -o /tmp/libuaf_trap.so -ldl
 *   LD_PRELOAD=/tmp/libuaf_trap.so python tools/soak.py ...
 * Environment: UAF_TRAP_MAX_BLOCK (bytes, default 65536: larger blocks are freed at once),
 *              UAF_TRAP_SLOTS (quarantine length, default 262144).
 */
#define _GNU_SOURCE
#include <execinfo.h>
#include <malloc.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

extern void __libc_free(void *);
extern void *__libc_malloc(size_t);

#define FRAMES 12
struct parked {
	void *p;
	size_t n;
	void *bt[FRAMES];
	int depth;
};

static struct parked *ring;
static size_t slots = 262144, head, count, max_block = 65536;
static pthread_mutex_t lock = PTHREAD_MUTEX_INITIALIZER;
static __thread int inside;
static int ready, reports;

static void report(const struct parked *q, size_t off) {
	char line[256];
	const unsigned char *b = (const unsigned char *)q->p;
	int n = snprintf(line, sizeof line, "\n[uaf_trap] block %p (%zu bytes) was written after free(), first change at offset %zu:", q->p, q->n, off);
	for (size_t k = off; k < q->n && k < off + 16; k++) n += snprintf(line + n, sizeof line - (size_t)n, " %02x", b[k]);
	n += snprintf(line + n, sizeof line - (size_t)n, "\n[uaf_trap] it had been freed by:\n");
	if (write(2, line, (size_t)n) < 0) {}
	backtrace_symbols_fd(q->bt, q->depth, 2);
	reports++;
}

static void verify(const struct parked *q) {
	const unsigned char *b = (const unsigned char *)q->p;
	for (size_t k = 0; k < q->n; k++)
		if (b[k] != 0xA5) {
			if (reports < 20) report(q, k);
			return;
		}
}

static void at_exit(void) {
	inside = 1;
	pthread_mutex_lock(&lock);
	for (size_t k = 0; k < count; k++) verify(&ring[(head + slots - count + k) % slots]);
	pthread_mutex_unlock(&lock);
	char line[96];
	int n = snprintf(line, sizeof line, "[uaf_trap] exit: %d block(s) written after free\n", reports);
	if (write(2, line, (size_t)n) < 0) {}
}

__attribute__((constructor)) static void init(void) {
	inside = 1;
	if (getenv("UAF_TRAP_MAX_BLOCK")) max_block = strtoull(getenv("UAF_TRAP_MAX_BLOCK"), 0, 10);
	if (getenv("UAF_TRAP_SLOTS")) slots = strtoull(getenv("UAF_TRAP_SLOTS"), 0, 10);
	ring = (struct parked *)__libc_malloc(slots * sizeof(struct parked));
	void *warm[4];
	backtrace(warm, 4); /* loads libgcc now, not inside a free() */
	atexit(at_exit);
	ready = ring != 0;
	inside = 0;
}

void free(void *p) {
	if (!p) return;
	if (!ready || inside) {
		__libc_free(p);
		return;
	}
	const size_t n = malloc_usable_size(p);
	if (n == 0 || n > max_block) {
		__libc_free(p);
		return;
	}
	inside = 1;
	struct parked in, out;
	in.p = p;
	in.n = n;
	in.depth = backtrace(in.bt, FRAMES);
	memset(p, 0xA5, n);
	int evict = 0;
	pthread_mutex_lock(&lock);
	if (count == slots) {
		out = ring[head];
		evict = 1;
	} else
		count++;
	ring[head] = in;
	head = (head + 1) % slots;
	pthread_mutex_unlock(&lock);
	if (evict) {
		verify(&out);
		__libc_free(out.p);
	}
	inside = 0;
}
