/* native_backtrace.c — TEST INFRASTRUCTURE: a native stack trace on a fatal signal.
 *
 * Python's faulthandler prints the PYTHON frames of a crashing process.  Round 3's one host crash (gpurun_out/r3a/tests.log:
 * "Fatal Python error: Segmentation fault", top frame the test function itself) left exactly that and nothing about the
 * native frame that faulted — a destructor, the library, the runtime — so its cause could only be narrowed down, never read
 * off.  tests/conftest.py builds this file and installs the handler for the whole session: on SIGSEGV / SIGBUS / SIGABRT /
 * SIGFPE / SIGILL it writes the native frames (module + offset: the in-tree .so files are reproducible builds, addr2line
 * resolves them) to stderr and then hands the signal to the handler that was there before (faulthandler's), so the
 * Python traceback follows as usual.  Async-signal-safe calls only (backtrace_symbols_fd writes straight to the fd). */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static struct sigaction g_prev[NSIG];
static int g_fd = 2;
static const int g_signals[] = {SIGSEGV, SIGBUS, SIGABRT, SIGFPE, SIGILL};

static volatile sig_atomic_t g_inside = 0;

static void on_fatal(int sig, siginfo_t* info, void* uc) {
    /* A fault INSIDE this handler — unwinding a smashed stack, which is the very case it exists for — must not come back
     * here: the second entry restores the default action and re-raises, so the process dies with its signal instead of
     * spinning or running the alternate stack over.  (The signal is also blocked while the handler runs: no SA_NODEFER.) */
    if (g_inside) {
        signal(sig, SIG_DFL);
        raise(sig);
        return;
    }
    g_inside = 1;
    static const char head[] = "\n=== native backtrace (tests/native_backtrace.c) ===\n";
    static const char tail[] = "=== end of native backtrace ===\n";
    void* frames[64];
    (void)!write(g_fd, head, sizeof head - 1);
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, g_fd);
    (void)!write(g_fd, tail, sizeof tail - 1);
    /* hand over to whoever was installed before us (faulthandler), or die of the default action */
    struct sigaction* prev = &g_prev[sig];
    if ((prev->sa_flags & SA_SIGINFO) && prev->sa_sigaction) prev->sa_sigaction(sig, info, uc);
    else if (prev->sa_handler != SIG_DFL && prev->sa_handler != SIG_IGN && prev->sa_handler) prev->sa_handler(sig);
    else {
        signal(sig, SIG_DFL);
        raise(sig);
    }
}

/* fd: where to write (a duplicate of the real stderr: the test runner redirects descriptor 2 while tests run) */
int rsv_test_install_native_backtrace(int fd) {
    if (fd >= 0) g_fd = fd;
    void* warm[4];
    (void)backtrace(warm, 4); /* loads libgcc now, not inside the handler */
    for (unsigned k = 0; k < sizeof g_signals / sizeof g_signals[0]; k++) {
        struct sigaction sa;
        memset(&sa, 0, sizeof sa);
        sa.sa_sigaction = on_fatal;
        sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
        sigemptyset(&sa.sa_mask);
        if (sigaction(g_signals[k], &sa, &g_prev[g_signals[k]]) != 0) return -1;
    }
    return 0;
}
