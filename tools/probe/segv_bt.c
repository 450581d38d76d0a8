// LD_PRELOAD shim: backtrace of a SIGSEGV / SIGABRT to stderr (there is no debugger in the image).
//   gcc -shared -fPIC -o /tmp/segv_bt.so tools/probe/segv_bt.c
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
static void handler(int sig, siginfo_t* si, void* ctx) {
    void* bt[64];
    (void)ctx;
    char msg[96];
    int n = snprintf(msg, sizeof msg, "\n== signal %d at address %p ==\n", sig, si ? si->si_addr : 0);
    if (write(2, msg, n) < 0) {}
    n = backtrace(bt, 64);
    backtrace_symbols_fd(bt, n, 2);
    _exit(128 + sig);
}
__attribute__((constructor)) static void install(void) {
    struct sigaction sa;
    sa.sa_sigaction = handler;
    sigemptyset(&sa.sa_mask);
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
    static char stack[1 << 16];
    stack_t ss = {stack, 0, sizeof stack};
    sigaltstack(&ss, 0);
    sigaction(SIGSEGV, &sa, 0);
    sigaction(SIGBUS, &sa, 0);
}
