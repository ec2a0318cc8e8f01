/* tools/segv_trace.c -- developer aid: LD_PRELOAD this to get a native backtrace (with module offsets) on
 * SIGSEGV / SIGBUS / SIGABRT from a process that loads the libraries through Python.
 *   gcc -O1 -g -shared -fPIC tools/segv_trace.c -o /tmp/libsegv_trace.so
 *   LD_PRELOAD=/tmp/libsegv_trace.so python tools/soak.py ...                                              */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

static void on_fault(int sig, siginfo_t *si, void *uc) {
	(void)uc;
	char head[128];
	int n = snprintf(head, sizeof head, "\n[segv_trace] signal %d, fault address %p, thread backtrace:\n", sig, si ? si->si_addr : 0);
	if (write(2, head, (size_t)n) < 0) {}
	void *frames[64];
	int depth = backtrace(frames, 64);
	backtrace_symbols_fd(frames, depth, 2);
	signal(sig, SIG_DFL);
	raise(sig);
}

__attribute__((constructor)) static void install(void) {
	struct sigaction sa;
	memset(&sa, 0, sizeof sa);
	sa.sa_sigaction = on_fault;
	sa.sa_flags = SA_SIGINFO | SA_NODEFER;
	sigaction(SIGSEGV, &sa, 0);
	sigaction(SIGBUS, &sa, 0);
	sigaction(SIGABRT, &sa, 0);
}
