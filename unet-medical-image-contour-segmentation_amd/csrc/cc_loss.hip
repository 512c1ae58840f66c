// cc_loss.hip -- host side of connected_component_loss (utils/connected_component_loss.py:7-60).
//
// The reference thresholds the probabilities on the device, copies each mask to the host and calls OpenCV
// (opencv-python~=4.11, not vendored, absent from this image): cv2.findContours(RETR_EXTERNAL,
// CHAIN_APPROX_SIMPLE) -> cv2.contourArea -> cv2.boundingRect.  This file restates the published algorithm:
// Suzuki-Abe border following (8-connected foreground) restricted to outermost borders; contourArea = |shoelace|
// of the traced closed chain through pixel centres (CHAIN_APPROX_SIMPLE only drops collinear points, the area is
// unchanged); boundingRect = the chain's integer bounding box.  PARITY UNPINNED: no OpenCV here to compare with;
// tests hold hand-derived known answers (single pixel -> 0, filled w x h rectangle -> (w-1)(h-1), ...).
// The loss is a host float without gradient, exactly as in the reference.
#include "uh_common.h"
#include <vector>
#include <cstdlib>
#include <cmath>

namespace {

struct Contour { double area; int x0, y0, x1, y1; };

// 8-neighbourhood in clockwise order starting at West: W, NW, N, NE, E, SE, S, SW
const int DX[8] = {-1, -1, 0, 1, 1, 1, 0, -1};
const int DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

void external_contours(const uint8_t* img, int H, int W, std::vector<Contour>& out) {
    const int PW = W + 2, PH = H + 2;
    std::vector<uint8_t> f((size_t)PW * PH, 0);      // padded foreground
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) f[(size_t)(y + 1) * PW + x + 1] = img[(size_t)y * W + x] ? 1 : 0;
    // outside = background 4-connected to the frame
    std::vector<uint8_t> outside((size_t)PW * PH, 0);
    std::vector<int> stack;
    stack.push_back(0);
    outside[0] = 1;
    while (!stack.empty()) {
        int p = stack.back();
        stack.pop_back();
        int y = p / PW, x = p - y * PW;
        const int nx[4] = {x - 1, x + 1, x, x}, ny[4] = {y, y, y - 1, y + 1};
        for (int k = 0; k < 4; ++k) {
            if (nx[k] < 0 || nx[k] >= PW || ny[k] < 0 || ny[k] >= PH) continue;
            int q = ny[k] * PW + nx[k];
            if (!f[q] && !outside[q]) { outside[q] = 1; stack.push_back(q); }
        }
    }
    std::vector<uint8_t> traced((size_t)PW * PH, 0);
    for (int y = 1; y <= H; ++y)
        for (int x = 1; x <= W; ++x) {
            const int p = y * PW + x;
            if (!f[p] || traced[p] || !outside[p - 1]) continue;
            // (x, y) starts an outermost outer border: trace it (Suzuki-Abe step 3, 8-connectivity)
            Contour c;
            c.x0 = c.x1 = x; c.y0 = c.y1 = y;
            double twice_area = 0.0;
            // find the first foreground neighbour, searching clockwise from West
            int d0 = -1;
            for (int k = 0; k < 8; ++k) {
                int q = (y + DY[k]) * PW + x + DX[k];
                if (f[q]) { d0 = k; break; }
            }
            traced[p] = 1;
            if (d0 < 0) { c.area = 0.0; out.push_back(c); continue; }     // isolated pixel
            // the traversal: at the current pixel, coming from direction `from` (index of the previous pixel among
            // the neighbours), search counter-clockwise starting just after `from` for the next foreground pixel
            int sx = x, sy = y;                      // start
            int px = x + DX[d0], py = y + DY[d0];    // i1j1 in Suzuki's notation: the pixel found clockwise
            int cx = sx, cy = sy;                    // current (i3, j3)
            int prevx = px, prevy = py;              // (i2, j2)
            for (;;) {
                // direction index of prev relative to current
                int dprev = 0;
                for (int k = 0; k < 8; ++k)
                    if (cx + DX[k] == prevx && cy + DY[k] == prevy) { dprev = k; break; }
                int nxp = -1, nyp = -1;
                for (int step = 1; step <= 8; ++step) {
                    int k = (dprev + 8 - step) & 7;              // counter-clockwise from the previous pixel
                    int qx = cx + DX[k], qy = cy + DY[k];
                    if (f[qy * PW + qx]) { nxp = qx; nyp = qy; break; }
                }
                traced[cy * PW + cx] = 1;
                // accumulate the edge current -> next for the shoelace formula
                twice_area += (double)cx * nyp - (double)nxp * cy;
                if (nxp < c.x0) c.x0 = nxp; if (nxp > c.x1) c.x1 = nxp;
                if (nyp < c.y0) c.y0 = nyp; if (nyp > c.y1) c.y1 = nyp;
                // termination (Suzuki 3.5): next == start and current == the first neighbour found
                if (nxp == sx && nyp == sy && cx == px && cy == py) break;
                prevx = cx; prevy = cy;
                cx = nxp; cy = nyp;
            }
            c.area = std::fabs(twice_area) * 0.5;
            // back to unpadded coordinates
            c.x0 -= 1; c.x1 -= 1; c.y0 -= 1; c.y1 -= 1;
            out.push_back(c);
        }
}

}  // namespace

// masks: HOST uint8 [B][H][W] (non-zero = foreground = p > 0.5, connected_component_loss.py:25).
// out[0] = sum of penalties / B  (the caller multiplies by penalty_weight, :59); out[1] = number of contours.
extern "C" int uh_cc_loss_host(const uint8_t* masks, int B, int H, int W, int edge_distance, int min_area, double* out) {
    UH_REQUIRE(masks && out && B > 0 && H > 0 && W > 0 && edge_distance > 0 && min_area > 0, "uh_cc_loss_host: bad args");
    double penalty = 0.0;
    size_t ncont = 0;
    std::vector<Contour> cs;
    for (int b = 0; b < B; ++b) {
        cs.clear();
        external_contours(masks + (size_t)b * H * W, H, W, cs);
        ncont += cs.size();
        for (const Contour& c : cs) {
            if (c.area < (double)min_area) {                       // :37-41 small component
                penalty += 1.0 - c.area / (double)min_area;
                continue;
            }
            const int wc = c.x1 - c.x0 + 1, hc = c.y1 - c.y0 + 1;  // cv2.boundingRect
            const int cx = c.x0 + wc / 2, cy = c.y0 + hc / 2;      // :45-46 integer division
            int d = cx;
            if (W - cx < d) d = W - cx;
            if (cy < d) d = cy;
            if (H - cy < d) d = H - cy;                            // :49-51
            if (d < edge_distance) penalty += 1.0 - (double)d / (double)edge_distance;   // :53-56
        }
    }
    out[0] = penalty / (double)B;
    out[1] = (double)ncont;
    return UH_OK;
}
