// Diagnostic preload: print a native backtrace of the faulting thread on SIGSEGV / SIGABRT (the GPU box has no gdb).
//   gcc -shared -fPIC -O1 -o /tmp/segv_bt.so tools/micro/segv_bt.c && LD_PRELOAD=/tmp/segv_bt.so python3 ...
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

static void on_fault(int sig, siginfo_t* si, void* ctx) {
  (void)ctx;
  void* frames[64];
  char head[96];
  int n = snprintf(head, sizeof head, "\n[segv_bt] signal %d at address %p, native frames of the faulting thread:\n", sig, si ? si->si_addr : (void*)0);
  if (write(2, head, (size_t)n) < 0) {}
  int k = backtrace(frames, 64);
  backtrace_symbols_fd(frames, k, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}

__attribute__((constructor)) static void install(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = on_fault;
  sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_RESETHAND;
  sigaction(SIGSEGV, &sa, 0);
  sigaction(SIGBUS, &sa, 0);
}
