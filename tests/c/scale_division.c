/* Exhaustive check of the encoder's division-free x = s / 32767 (sela_encode.hip: scale_sample):
 * q0 = s * RN(1/32767); e = fma(-32767, q0, s); q = fma(e, RN(1/32767), q0) must equal the IEEE
 * quotient bit for bit for every s the encoder can see (16-bit samples and their differences).
 * Prints the number of mismatches.  Build: gcc -O2 -ffp-contract=off scale_division.c -lm */
#include <math.h>
#include <stdio.h>
#include <string.h>

int main(void)
{
    const double d = 32767.0, r = 1.0 / 32767.0;
    long bad = 0;
    for (long n = -70000; n <= 70000; n++) {
        const double x = (double)n, ref = x / d;
        const double q0 = x * r;
        const double e = fma(-d, q0, x);
        const double q = fma(e, r, q0);
        bad += memcmp(&q, &ref, sizeof q) != 0;
    }
    printf("%ld\n", bad);
    return 0;
}
