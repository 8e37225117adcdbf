// (MT, NW, NT) tile configurations instantiated for every epilogue mode and vector width
#pragma once
#define X3_CONFIGS X3_CASES(2, 8, 2) X3_CASES(2, 4, 4) X3_CASES(2, 8, 1) X3_CASES(1, 8, 2) X3_CASES(1, 8, 1)
