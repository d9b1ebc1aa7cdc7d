/* A plain C99 consumer of include/icpflow_hip.h: what a non-Python host (cgo / JNI / FFI) sees.
 * Built and run by tests/test_abi_exports.py with gcc; touches no GPU: versions, argument errors as status
 * codes, and the host-side HDBSCAN tree function on a 6-point path graph. */
#include <stdio.h>
#include <string.h>

#include "icpflow_hip.h"

int main(void)
{
    if (icpflow_version() != ICPFLOW_VERSION) return 1;
    if (strlen(icpflow_build_info()) == 0) return 1;
    /* per-call options are validated before anything else: a struct from another header version is refused */
    icpflow_options_t opt;
    memset(&opt, 0, sizeof(opt));
    opt.struct_size = sizeof(opt) - 4;
    int rco = icpflow_icp(NULL, NULL, NULL, 1, 1, 0.1, 10, 1e-6, ICPFLOW_STOP_REFERENCE, NULL, NULL, NULL, NULL, NULL,
                          NULL, 0, NULL, &opt);
    if (rco != ICPFLOW_E_ARG || strstr(icpflow_last_error(), "struct_size") == NULL) return 10;
    opt.struct_size = sizeof(opt);
    opt.icp_search = 9;
    rco = icpflow_icp(NULL, NULL, NULL, 1, 1, 0.1, 10, 1e-6, ICPFLOW_STOP_REFERENCE, NULL, NULL, NULL, NULL, NULL, NULL,
                      0, NULL, &opt);
    if (rco != ICPFLOW_E_ARG || strstr(icpflow_last_error(), "icp_search") == NULL) return 11;
    if (icpflow_workspace_bytes(0, 10, 0, 0, 0) != 0) return 2;
    int rc = icpflow_dbscan(NULL, 3, NULL, 10, 0.25, 20, NULL, NULL, NULL, NULL, 0, NULL);
    if (rc != ICPFLOW_E_ARG || strstr(icpflow_last_error(), "null pointer") == NULL) return 3;
    float pts[3] = {0.f, 0.f, 0.f};
    int32_t out[8];
    rc = icpflow_dbscan(pts, 3, NULL, 1, -1.0, 20, out, out, out, out, 0, NULL);
    if (rc != ICPFLOW_E_ARG || strstr(icpflow_last_error(), "eps") == NULL) return 4;
    double w2[1];
    rc = icpflow_hdbscan_mst(pts, 3, NULL, 1, 100, 0.25, NULL, out, out, w2, out, out, out, 0, NULL);
    if (rc != ICPFLOW_E_LIMIT) return 5;
    /* two tight triples joined by one long edge, min_cluster_size 3 -> two clusters */
    const int32_t a[5] = {0, 1, 3, 4, 2}, b[5] = {1, 2, 4, 5, 3};
    const double w[5] = {0.10, 0.11, 0.12, 0.13, 5.0};
    int32_t labels[6];
    rc = icpflow_hdbscan_labels(a, b, w, 6, 3, labels);
    if (rc != 0) return 6;
    if (!(labels[0] == labels[1] && labels[1] == labels[2] && labels[3] == labels[4] && labels[4] == labels[5])) return 7;
    if (labels[0] == labels[3] || labels[0] < 0 || labels[3] < 0) return 8;
    if (icpflow_hdbscan_labels(a, b, w, 7, 3, labels) == 0) return 9;   /* five edges cannot span seven points */
    printf("ok %d %d\n", labels[0], labels[3]);
    return 0;
}
