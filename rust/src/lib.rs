//! `orc_format::read::decode` iterators backed by the MI355X HIP decoder (liborcd.so).
//!
//! Same type names and constructor signatures as the crate (src/read/decode/mod.rs:8-14), so that
//! `use orc_format::read::decode::*` can be pointed here.  Each iterator drains its reader (one
//! stream of one column, as Column::get_stream hands it out), makes ONE call into the C ABI, and
//! then iterates the decoded buffer.
//!
//! UNCOMPILED: this build environment has no Rust toolchain; the file documents the binding a
//! maintainer of the crate would add and has never been through rustc.
mod ffi;

use std::io::Read;
use std::os::raw::c_int;

#[derive(Debug, Clone)]
pub enum Error {
    OutOfSpec,
    InvalidUtf8,
    InvalidColumn(u32),
    InvalidKind(u32),
    DecodeFloat,
    Decompression,
    InvalidProto,
    Unsupported,
    Device,
}

impl Error {
    fn from_status(status: c_int) -> Self {
        match status {
            ffi::ORCD_OUT_OF_SPEC => Error::OutOfSpec,
            ffi::ORCD_INVALID_UTF8 => Error::InvalidUtf8,
            ffi::ORCD_INVALID_COLUMN => Error::InvalidColumn(0),
            ffi::ORCD_INVALID_KIND => Error::InvalidKind(0),
            ffi::ORCD_DECODE_FLOAT => Error::DecodeFloat,
            ffi::ORCD_DECOMPRESSION => Error::Decompression,
            ffi::ORCD_INVALID_PROTO => Error::InvalidProto,
            ffi::ORCD_UNSUPPORTED => Error::Unsupported,
            _ => Error::Device,
        }
    }
}

thread_local! {
    // one decode context per thread (orcd.h: a context belongs to one thread and one GPU)
    static CTX: *mut ffi::orcd_ctx = {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { ffi::orcd_ctx_create(0, &mut ctx) };
        assert_eq!(rc, ffi::ORCD_OK, "orcd_ctx_create failed");
        ctx
    };
}

fn ctx() -> *mut ffi::orcd_ctx {
    CTX.with(|c| *c)
}

struct Decoded<T> {
    values: std::vec::IntoIter<T>,
    status: c_int,
}

impl<T> Decoded<T> {
    fn next(&mut self) -> Option<Result<T, Error>> {
        if self.status != ffi::ORCD_OK {
            let e = Error::from_status(self.status);
            self.status = ffi::ORCD_OK;
            self.values = Vec::new().into_iter();
            return Some(Err(e));
        }
        self.values.next().map(Ok)
    }
}

fn decode_rlev2<R: Read>(reader: &mut R, length: usize, scratch: &mut Vec<u8>, signed: bool) -> Decoded<i64> {
    scratch.clear();
    let mut values = vec![0i64; length];
    let status = match reader.read_to_end(scratch) {
        Err(_) => ffi::ORCD_OUT_OF_SPEC,
        Ok(_) => unsafe {
            ffi::orcd_rlev2_decode(ctx(), scratch.as_ptr(), scratch.len(), length, std::ptr::null(), 0,
                                   values.as_mut_ptr(), if signed { ffi::ORCD_SIGNED } else { 0 })
        },
    };
    Decoded { values: values.into_iter(), status }
}

/// SignedRleV2Iter::new(reader, length, scratch) -- src/read/decode/rle_v2.rs:713-758
pub struct SignedRleV2Iter<R: Read> {
    reader: R,
    scratch: Vec<u8>,
    decoded: Decoded<i64>,
}

impl<R: Read> SignedRleV2Iter<R> {
    pub fn new(mut reader: R, length: usize, mut scratch: Vec<u8>) -> Self {
        let decoded = decode_rlev2(&mut reader, length, &mut scratch, true);
        Self { reader, scratch, decoded }
    }
    pub fn into_inner(mut self) -> (R, Vec<u8>) {
        self.scratch.clear();
        (self.reader, self.scratch)
    }
}

impl<R: Read> Iterator for SignedRleV2Iter<R> {
    type Item = Result<i64, Error>;
    fn next(&mut self) -> Option<Self::Item> {
        self.decoded.next()
    }
}

/// UnsignedRleV2Iter::new(reader, length, scratch) -- src/read/decode/rle_v2.rs:517-562
pub struct UnsignedRleV2Iter<R: Read> {
    reader: R,
    scratch: Vec<u8>,
    decoded: Decoded<i64>,
}

impl<R: Read> UnsignedRleV2Iter<R> {
    pub fn new(mut reader: R, length: usize, mut scratch: Vec<u8>) -> Self {
        let decoded = decode_rlev2(&mut reader, length, &mut scratch, false);
        Self { reader, scratch, decoded }
    }
    pub fn into_inner(mut self) -> (R, Vec<u8>) {
        self.scratch.clear();
        (self.reader, self.scratch)
    }
}

impl<R: Read> Iterator for UnsignedRleV2Iter<R> {
    type Item = Result<u64, Error>;
    fn next(&mut self) -> Option<Self::Item> {
        self.decoded.next().map(|r| r.map(|v| v as u64))
    }
}

/// BooleanIter::new(reader, length) -- src/read/decode/boolean_rle.rs:65-158
pub struct BooleanIter<R: Read> {
    reader: R,
    decoded: Decoded<u8>,
}

impl<R: Read> BooleanIter<R> {
    pub fn new(mut reader: R, length: usize) -> Self {
        let mut stream = Vec::new();
        let mut values = vec![0u8; length];
        let status = match reader.read_to_end(&mut stream) {
            Err(_) => ffi::ORCD_OUT_OF_SPEC,
            Ok(_) => unsafe {
                ffi::orcd_boolean_decode(ctx(), stream.as_ptr(), stream.len(), length, std::ptr::null(), 0,
                                         values.as_mut_ptr(), std::ptr::null_mut(), std::ptr::null_mut(), 0)
            },
        };
        Self { reader, decoded: Decoded { values: values.into_iter(), status } }
    }
    pub fn into_inner(self) -> R {
        self.reader
    }
}

impl<R: Read> Iterator for BooleanIter<R> {
    type Item = Result<bool, Error>;
    fn next(&mut self) -> Option<Self::Item> {
        self.decoded.next().map(|r| r.map(|b| b != 0))
    }
}

/// FloatIter::<f32 | f64, R>::new(reader, length) -- src/read/decode/float.rs:34-91
pub struct FloatIter<T: Copy + Default, R: Read> {
    reader: R,
    decoded: Decoded<T>,
}

impl<T: Copy + Default, R: Read> FloatIter<T, R> {
    pub fn new(mut reader: R, length: usize) -> Self {
        let mut stream = Vec::new();
        let mut values = vec![T::default(); length];
        let status = match reader.read_to_end(&mut stream) {
            Err(_) => ffi::ORCD_DECODE_FLOAT,
            Ok(_) => unsafe {
                ffi::orcd_float_decode(ctx(), stream.as_ptr(), stream.len(), std::mem::size_of::<T>(), length,
                                       values.as_mut_ptr() as *mut _, 0)
            },
        };
        Self { reader, decoded: Decoded { values: values.into_iter(), status } }
    }
    pub fn into_inner(self) -> R {
        self.reader
    }
}

impl<T: Copy + Default, R: Read> Iterator for FloatIter<T, R> {
    type Item = Result<T, Error>;
    fn next(&mut self) -> Option<Self::Item> {
        self.decoded.next()
    }
}
