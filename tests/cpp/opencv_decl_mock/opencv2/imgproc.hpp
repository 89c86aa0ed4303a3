// SYNTAX-CHECK MOCK - see core.hpp in this directory.  Declarations only, from OpenCV's public imgproc API (drawing functions).
#ifndef FRT_TEST_OPENCV_DECL_MOCK_IMGPROC_HPP
#define FRT_TEST_OPENCV_DECL_MOCK_IMGPROC_HPP
#include "core.hpp"
namespace cv {
enum HersheyFonts { FONT_HERSHEY_SIMPLEX = 0, FONT_HERSHEY_PLAIN = 1, FONT_HERSHEY_DUPLEX = 2 };
enum LineTypes { LINE_8 = 8 };
void rectangle(Mat &img, Point pt1, Point pt2, const Scalar &color, int thickness = 1, int lineType = LINE_8, int shift = 0);
void putText(Mat &img, const String &text, Point org, int fontFace, double fontScale, Scalar color, int thickness = 1, int lineType = LINE_8,
             bool bottomLeftOrigin = false);
}  // namespace cv
#endif
