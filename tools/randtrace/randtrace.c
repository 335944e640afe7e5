// LD_PRELOAD shim: who calls libc rand() / srand() in a process?  Counts calls per (calling shared object, thread) and prints the table at exit.
// gcc -O2 -shared -fPIC -o librandtrace.so randtrace.c -ldl ;  LD_PRELOAD=./librandtrace.so python -m pytest ...
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <unistd.h>
#include <sys/syscall.h>
static int (*real_rand)(void); static void (*real_srand)(unsigned);
static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
struct Row { char lib[160]; long tid; long rand_calls, srand_calls; };
static struct Row rows[256]; static int nrows;
static void note(void* ra, int is_srand)
{
	Dl_info di; const char* name = "?";
	if (dladdr(ra, &di) && di.dli_fname) name = di.dli_fname;
	long tid = syscall(SYS_gettid);
	pthread_mutex_lock(&mu);
	int i; for (i = 0; i < nrows; i++) if (rows[i].tid == tid && !strncmp(rows[i].lib, name, 159)) break;
	if (i == nrows && nrows < 256) { strncpy(rows[i].lib, name, 159); rows[i].tid = tid; rows[i].rand_calls = rows[i].srand_calls = 0; nrows++; }
	if (i < 256) { if (is_srand) rows[i].srand_calls++; else rows[i].rand_calls++; }
	pthread_mutex_unlock(&mu);
}
int rand(void) { if (!real_rand) real_rand = dlsym(RTLD_NEXT, "rand"); note(__builtin_return_address(0), 0); return real_rand(); }
void srand(unsigned s) { if (!real_srand) real_srand = dlsym(RTLD_NEXT, "srand"); note(__builtin_return_address(0), 1); real_srand(s); }
__attribute__((destructor)) static void dump(void)
{
	const char* out = getenv("RANDTRACE_OUT"); FILE* f = out ? fopen(out, "a") : stderr; if (!f) f = stderr;
	fprintf(f, "# rand()/srand() callers of pid %d (main thread %d)\n", (int)getpid(), (int)getpid());
	for (int i = 0; i < nrows; i++) fprintf(f, "%-90s tid %-8ld rand %-10ld srand %ld\n", rows[i].lib, rows[i].tid, rows[i].rand_calls, rows[i].srand_calls);
	if (f != stderr) fclose(f);
}
