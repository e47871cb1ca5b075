/* gkm_host.h -- internal declarations shared by the host C files. */
#ifndef GKM_HOST_H
#define GKM_HOST_H

#include <stdint.h>
#include "../../include/gkmkern_pylib.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GKM_MAX_L 12      /* libgkm.h:31 MAX_MM / gkmkern_pylib.c:54 */
#define GKM_MAX_SEQ 2047  /* libgkm.h:32 MAX_SEQ_LENGTH - 1 */

enum { GKM_LOG_TRACE = 0, GKM_LOG_DEBUG, GKM_LOG_INFO, GKM_LOG_WARN, GKM_LOG_ERROR };

void gkm_log_set_level(int level);
int gkm_log_enabled(int level);
int gkm_log_level_from_verbosity(int verbosity); /* -1 if verbosity is not 0..4 */
void gkm_log(int level, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

const int64_t *gkm_problem_offsets(const gkm_problem *p);
const uint8_t *gkm_problem_all_codes(const gkm_problem *p);

#ifdef __cplusplus
}
#endif
#endif
