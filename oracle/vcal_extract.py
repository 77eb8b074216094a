#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  The vignetteCalib solver's accumulate loops live inside main() of the reference's
src/main_vignetteCalib.cpp (:395-527) -- they cannot be linked, and the file as a whole needs aruco + OpenCV.
This script cuts the loop bodies (and getInterpolatedElement, :52-70) out of the reference file WHERE IT LIES,
by their comment anchors (the output smoothing :541-566 by its first and last statement), into oracle/_ref/*.inc
(git-ignored, never committed); oracle/vcal_ref_wrapper.cpp wraps them into three functions and oracle/Makefile compiles that into oracle/_ref/libvcal_ref.so -- the pin for
oracle/mdc_oracle.c's restatement (tests/test_vcal.py).  usage: vcal_extract.py <reference src dir> <out dir>"""
import os
import sys

src, out = sys.argv[1], sys.argv[2]
lines = open(os.path.join(src, "main_vignetteCalib.cpp")).read().split("\n")


def find(sub, start=0):
    for i in range(start, len(lines)):
        if sub in lines[i]:
            return i
    raise SystemExit("anchor %r not found in main_vignetteCalib.cpp" % sub)


i0 = find("float getInterpolatedElement(")
i1 = next(i for i in range(i0, len(lines)) if lines[i].rstrip() == "}")
a0 = find("optimize planeColor")
b0 = find("optimize vignette", a0)
b1 = find("vignetteFactor[pi] /= maxFac;", b0)
# "dilate & smoothe vignette by 4 pixel for output" (:541-566): from the memcpy into TT to the end of the dilit loop
c0 = find("memcpy(vignetteFactorTT, vignetteFactor", b1)
c1 = find('displayImageV(vignetteFactorTT', c0)  # its own block opens on the line before
while lines[c1 - 1].strip() in ("{", ""):
    c1 -= 1
# plane points that fall outside the image -> NaN coordinates (:345-357)
d0 = find("int u_d = plane2imgX[x+y*gw]+0.5;") - 3
assert lines[d0].strip().startswith("for(int x=0; x<gw;x++)"), lines[d0]
d1 = find("cv::imshow(\"inRaw\",dbgImg);", d0)
# gradient mask of a calibration image (:293-301): the double loop up to images.push_back
e0 = find("for(int y=2; y<hI-2;y++)")
e1 = find("images.push_back(image);", e0)
os.makedirs(out, exist_ok=True)
open(os.path.join(out, "vcal_interp.inc"), "w").write("\n".join(lines[i0:i1 + 1]) + "\n")
open(os.path.join(out, "vcal_body_plane.inc"), "w").write("\n".join(lines[a0 + 1:b0]) + "\n")
open(os.path.join(out, "vcal_body_vignette.inc"), "w").write("\n".join(lines[b0 + 1:b1 + 1]) + "\n")
open(os.path.join(out, "vcal_body_smooth.inc"), "w").write("\n".join(lines[c0:c1]) + "\n")
open(os.path.join(out, "vcal_body_mask.inc"), "w").write("\n".join(lines[d0:d1]) + "\n")
open(os.path.join(out, "vcal_body_gradmask.inc"), "w").write("\n".join(lines[e0:e1]) + "\n")
print("vcal: interp %d-%d, plane step %d-%d, vignette step %d-%d, smoothing %d-%d" % (i0 + 1, i1 + 1, a0 + 2, b0, b0 + 2, b1 + 1, c0 + 1, c1))
