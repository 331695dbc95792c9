"""CPU oracle for the encode-and-contrast path: TEST INFRASTRUCTURE, never imported by the product."""
